#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 300 python3 scripts/ratio_sweep.py gpurun_out/r03_ratio_sweep.json 2>&1 | grep -v amdgpu
timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep -v amdgpu
timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep one_frame
timeout 200 python3 scripts/ratio_diag.py c1_frames:65536 pickled_frame:1024 text:1024 2>&1 | grep -v amdgpu
timeout 1500 python3 -m pytest tests -m gpu -q -x > gpurun_out/r03_pytest4.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03_pytest4.log
