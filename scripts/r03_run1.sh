#!/bin/bash
# round 3, first GPU trip: the GPU test suite, the small-size ratio sweep, and rocprof / PMC evidence for the DEFAULT
# (one-frame) layout's kernels
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest1.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03_pytest1.log
timeout 300 python3 scripts/ratio_sweep.py gpurun_out/r03_ratio_sweep.json > gpurun_out/r03_ratio_sweep.txt 2>&1; cat gpurun_out/r03_ratio_sweep.txt | grep -v amdgpu.ids
rm -rf /tmp/prof_of /tmp/pmc_of_f /tmp/pmc_of_w
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_of -- python3 scripts/one_frame_prof.py 1024 3 > gpurun_out/r03_one_frame_run.log 2>&1
grep "one frame" gpurun_out/r03_one_frame_run.log
cp "$(ls /tmp/prof_of/*/*kernel_stats.csv | head -1)" gpurun_out/r03_one_frame_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pmc_of_f -- python3 scripts/one_frame_prof.py 1024 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pmc_of_w -- python3 scripts/one_frame_prof.py 1024 2 > /dev/null 2>&1
python3 scripts/pmc_util.py /tmp/pmc_of_f > gpurun_out/r03_one_frame_pmc_fetch.txt 2>&1
python3 scripts/pmc_util.py /tmp/pmc_of_w > gpurun_out/r03_one_frame_pmc_write.txt 2>&1
head -40 gpurun_out/r03_one_frame_kernel_stats.csv | cut -c1-200
