#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_bigframe_gpu.py -m gpu -q -x 2>&1 | tail -15
rm -rf /tmp/prof_n1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n1 -- python3 scripts/dev/exp_n1.py 1024 > gpurun_out/n1_run.log 2>&1
cat gpurun_out/n1_run.log | grep -v amdgpu
cp "$(ls /tmp/prof_n1/*/*kernel_stats.csv | head -1)" gpurun_out/r03_n1_kernel_stats.csv 2>/dev/null
grep "zl" gpurun_out/r03_n1_kernel_stats.csv | cut -c1-170 | head -30
