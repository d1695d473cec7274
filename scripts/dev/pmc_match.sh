#!/bin/bash
# PMC passes over the match kernel.  usage: pmc_match.sh "<variant list: base = in-tree>" "<kind>" "<counter set 1>" "<counter set 2>" ...
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
variants=$1; kind=$2; shift 2
for v in $variants; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  i=0
  for set in "$@"; do
    i=$((i+1)); rm -rf /tmp/pmc_${v}_$i
    timeout 60 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc_${v}_$i -- python3 scripts/dev/pmc_match.py $kind > /tmp/pmc_log_${v}_$i.txt 2>&1 || tail -3 /tmp/pmc_log_${v}_$i.txt
    echo "== $v set $i"; python3 scripts/pmc_util.py /tmp/pmc_${v}_$i 2>/dev/null | grep -A12 "zl_match" | grep -v entropy
  done
done
