#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in base sub16 sub64; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== variant $v"
  timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py text 256 2>&1 | grep one_frame
  timeout 300 python3 scripts/ratio_sweep.py 2>&1 | grep -v amdgpu
done
unset ZL_HIP_LIB
timeout 300 python3 scripts/e2e_bench.py 1024 auto 2>&1 | grep -v amdgpu
