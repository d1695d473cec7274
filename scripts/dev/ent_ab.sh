#!/bin/bash
# Dev: the entropy stage as one kernel against two (and the second half at 4 / 5 / 8 workgroups per CU): same bytes? how fast?
ZL_HIP_LIB=zstdlite_amd/build/variants/libzl_hip.entone.so timeout 300 python scripts/dev/same_bytes.py write /tmp/sb_ent.json 2>&1 | tail -1
for v in "" entone ent2w5 ent2w8; do
  lib=zstdlite_amd/libzl_hip.so; [ -n "$v" ] && lib=zstdlite_amd/build/variants/libzl_hip.$v.so
  echo "== ${v:-default (two kernels, second at 4 per CU)}"
  ZL_HIP_LIB=$lib timeout 300 python scripts/dev/same_bytes.py check /tmp/sb_ent.json 2>&1 | tail -1
  ZL_HIP_LIB=$lib timeout 300 python bench.py --no-extra --no-cpu --no-c2 --steps 8 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('value', round(l['value'],1), {k: round(v,3) for k,v in l['kernel_ms'].items() if k in ('match','entropy','gather')})"
done
