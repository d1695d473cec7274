#!/bin/bash
# A/B of match-kernel variants inside one gpurun session: per-input kernel times without and with phase stamps.
# usage: scripts/dev/ring_ab.sh variant...   ("base" = the in-tree library)
for v in "$@"; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== $v"
  ZL_PROF=1 python scripts/kernel_bench.py 268435456 text,iid10,rand,int20 2>&1 | grep -v "amdgpu.ids\|phases\|roles" | sed 's/entropy .*//'
  ZL_PROF=1 python scripts/kernel_bench.py 939524096 text 2>&1 | grep -v "amdgpu.ids\|phases\|roles" | sed 's/entropy .*//'
  ZL_PROF=1 python scripts/kernel_bench.py 1073741824 mix 2>&1 | grep -v "amdgpu.ids\|phases\|roles" | sed 's/scan .*//'
  ZL_LAYOUT=one_frame ZL_PROF=1 python scripts/kernel_bench.py 1073741824 mix 2>&1 | grep -v "amdgpu.ids\|phases\|roles" | sed 's/scan .*far/far/; s/decode .*//; s/^/one_frame /'
  [ -n "$ZL_AB_STAMPS" ] && ZL_PROF=2 python scripts/kernel_bench.py 268435456 text 2>&1 | grep "match phases"
done
