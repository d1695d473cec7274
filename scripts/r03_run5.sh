#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep -v amdgpu
timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep one_frame
timeout 100 python3 scripts/layout_bench.py records 64 2>&1 | grep one_frame
for k in rand iid10 text int20; do timeout 100 python3 scripts/layout_bench.py $k 256 2>&1 | grep one_frame; done
timeout 300 python3 scripts/ratio_sweep.py gpurun_out/r03_ratio_sweep.json 2>&1 | grep -v amdgpu
timeout 1500 python3 -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest3.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03_pytest3.log
rm -rf /tmp/prof_of
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_of -- python3 scripts/one_frame_prof.py 1024 3 > /dev/null 2>&1
grep "far\|match\|entropy_kernel" "$(ls /tmp/prof_of/*/*kernel_stats.csv | head -1)" | cut -c1-160
