#!/bin/bash
# Runs one command once per library variant (A/B inside ONE gpurun session: boxes differ by 1-2 %).
# usage: scripts/ab.sh "<command>" variant1 variant2 ...   ("base" = the in-tree library)
cmd=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== $v"
  bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
done
