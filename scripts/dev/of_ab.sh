#!/bin/bash
# one-frame layout A/B: kernel times on the 1 GiB mix + ratios on the long-range inputs, per library variant
for v in "$@"; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== $v"
  ZL_LAYOUT=one_frame ZL_PROF=1 python scripts/kernel_bench.py 1073741824 mix 2>&1 | grep -v "amdgpu.ids\|phases\|roles" | sed 's/scan .*far/far/; s/decode .*//'
  ZL_LAYOUT=one_frame ZL_PROF=1 python scripts/kernel_bench.py 268435456 text 2>&1 | grep -v "amdgpu.ids\|phases\|roles" | sed 's/scan .*far/far/; s/decode .*//'
  python scripts/ratio_longrange.py 2>&1 | grep -v "amdgpu.ids\|^{" | cut -c1-150
done
