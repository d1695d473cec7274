#!/usr/bin/env python3
"""Dev: our ratio at a level against the reference's at the same level, on text and the corpus."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
inputs = {"text16M": GENERATORS["text"](16 << 20), "corpus": LONG_RANGE["corpus"](None)}
for lvl in (3, 5, 6, 9, 12):  # (6 and up: the operating point 6)
    for name, data in inputs.items():
        ours = len(z.zstd_compress(data, level=lvl))
        theirs = len(ref.compress(data, level=lvl))
        print(f"level {lvl:2d} {name:8s} ours {len(data)/ours:6.3f} ref {len(data)/theirs:6.3f} size ratio {ours/theirs:.3f}", flush=True)
