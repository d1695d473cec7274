#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in base b0 c40 s12 s11; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== variant $v"
  timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py text 256 2>&1 | grep one_frame
done
