#!/bin/bash
# One gpurun session: same-bytes check of the in-tree library against a reference variant, then the A/B timings.
# usage: scripts/dev/check_ab.sh <reference variant> variant...
ref=$1; shift
ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$ref.so python scripts/dev/same_bytes.py write /tmp/sb_ref.json 2>&1 | grep -v amdgpu.ids
python scripts/dev/same_bytes.py check /tmp/sb_ref.json 2>&1 | grep -v amdgpu.ids
scripts/dev/ring_ab.sh "$@"
