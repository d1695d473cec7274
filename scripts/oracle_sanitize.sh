#!/bin/bash
# oracle/zl_oracle.c under ASan + UBSan on intact and damaged frames (needs oracle/_ref for the frames; CPU only).
# usage: scripts/oracle_sanitize.sh
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o "$tmp/oracle_san" tests/c/oracle_san.c oracle/zl_oracle.c -Ioracle
python3 - "$tmp" <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from zl_testlib import GENERATORS, Ref
ref = Ref()
for i, (k, n, l) in enumerate([("text", 300000, 3), ("iid10", 200000, 3), ("int20", 150000, 1), ("text", 1500000, 19), ("rand", 70000, 3), ("text", 5000, 3), ("text", 700000, -5)]):
    open(f"{sys.argv[1]}/f{i}.zst", "wb").write(ref.compress(GENERATORS[k](n, seed=i), level=l, checksum=bool(i & 1)))
PY
"$tmp/oracle_san" "$tmp"/f*.zst tests/golden/data.json.zst
rm -rf "$tmp"
echo "oracle: clean under ASan + UBSan"
