#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in base xcd0 f65 f80 f50; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  echo "=== variant $v"
  timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py records 64 2>&1 | grep one_frame
  timeout 100 python3 scripts/layout_bench.py text 256 2>&1 | grep one_frame
done
unset ZL_HIP_LIB
rm -rf /tmp/prof_of
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_of -- python3 scripts/one_frame_prof.py 1024 3 > /dev/null 2>&1
grep "far\|match" "$(ls /tmp/prof_of/*/*kernel_stats.csv | head -1)" | cut -c1-160
