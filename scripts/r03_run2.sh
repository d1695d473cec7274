#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r03_pytest2.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03_pytest2.log
timeout 200 python3 scripts/ratio_diag.py f64_seq:65536 f64_seq:1048576 c1_frames:65536 c1_frames:1048576 text:1024 pickled_frame:1024 f64_rounded:1024 pickled_frame:4096 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_ratio_diag.txt; cat gpurun_out/r03_ratio_diag.txt
for k in rand iid10 text int20; do timeout 100 python3 scripts/layout_bench.py $k 256 2>&1 | grep -v amdgpu.ids; done
for v in base s4 d16 d24; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== variant $v"
  timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py records 64 2>&1 | grep one_frame
done
