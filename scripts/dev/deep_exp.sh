#!/bin/bash
# Dev: what bounds zl_deep_search_kernel -- variants of the library with parts of the kernel taken out (ZL_DEEP_EXP bits: 1 no
# scattered store, 2 no record loads, 4 the store coalesced), the search stage timed on 256 MiB of text and on the mix.
for v in 0 1 2 3 4; do
  python -m zstdlite_amd.build --variant exp$v -DZL_DEEP_EXP=$v > /dev/null 2>&1
  echo "== ZL_DEEP_EXP=$v"
  ZL_HIP_LIB=zstdlite_amd/build/variants/libzl_hip.exp$v.so python scripts/dev/deep_speed.py 9 2>&1 | grep -v amdgpu.ids | grep "one_frame"
done
