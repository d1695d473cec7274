#!/usr/bin/env python3
"""Dev: the price-based parse of levels 16 - 22 (zl_opt.hip) against the reference at the same level: round trip through the REFERENCE
decoder and ours, compressed size next to the reference's and next to level 15's, wall time of the host call."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
corpus = LONG_RANGE["corpus"](None)
inputs = {"corpus": corpus, "text4M": GENERATORS["text"](4 << 20), "records8M": LONG_RANGE["records"](8 << 20), "iid1M": GENERATORS["iid10"](1 << 20),
          "int20_1M": GENERATORS["int20"](1 << 20), "rand300K": GENERATORS["rand"](300_000), "corpus1M": corpus[3_000_000:4_000_000], "tiny": corpus[:3000], "seven": b"abcdefg", "empty": b""}
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [15, 16, 19, 22]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
bad = 0
for name, data in inputs.items():
    if only and name not in only:
        continue
    for layout in ("auto", "frames") if name in ("corpus", "corpus1M") else ("auto",):
        for lvl in levels:
            cctx = z.zstd_cctx(level=lvl, frame_layout=layout)
            t0 = time.time(); comp = z.zstd_compress(data, cctx=cctx); dt = time.time() - t0
            t0 = time.time(); comp2 = z.zstd_compress(data, cctx=cctx); dt = min(dt, time.time() - t0)
            back = ref.decompress(comp, len(data)) if data else b""
            ok = back == data and z.zstd_decompress(comp) == data and comp == comp2
            bad += not ok
            theirs = len(ref.compress(data, level=lvl))
            print(f"{name:9s} {layout:6s} level {lvl:2d} ours {len(comp):9d} ref {theirs:9d} size ratio {len(comp)/max(theirs,1):.3f} {'ok' if ok else 'MISMATCH'} {dt*1e3:8.1f} ms host call", flush=True)
print("FAILED" if bad else "all round trips ok")
sys.exit(1 if bad else 0)
