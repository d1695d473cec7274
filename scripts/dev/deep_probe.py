#!/usr/bin/env python3
"""Dev: the deep match finder of levels >= 6 (zl_deep.hip) against the reference at the same level: round trip through the
REFERENCE decoder, compressed size next to the reference's, wall time of the host call."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
inputs = {"corpus": LONG_RANGE["corpus"](None), "text16M": GENERATORS["text"](16 << 20), "records8M": LONG_RANGE["records"](8 << 20),
          "text200K": GENERATORS["text"](200_000), "corpus1M": LONG_RANGE["corpus"](None)[3_000_000:4_000_000], "tiny": LONG_RANGE["corpus"](None)[:3000]}
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5, 6, 7, 9, 12, 15]
bad = 0
for name, data in inputs.items():
    for layout in ("auto", "frames"):
        for lvl in levels:
            cctx = z.zstd_cctx(level=lvl, frame_layout=layout)
            t0 = time.time(); comp = z.zstd_compress(data, cctx=cctx); dt = time.time() - t0
            t0 = time.time(); comp = z.zstd_compress(data, cctx=cctx); dt = min(dt, time.time() - t0)
            back = ref.decompress(comp, len(data))
            ok = back == data and z.zstd_decompress(comp) == data
            bad += not ok
            theirs = len(ref.compress(data, level=lvl))
            print(f"{name:9s} {layout:6s} level {lvl:2d} ours {len(data)/len(comp):7.3f} ref {len(data)/theirs:7.3f} size ratio {len(comp)/theirs:.3f} {'ok' if ok else 'MISMATCH'} {len(data)/dt/1e9:6.2f} GB/s host call", flush=True)
print("FAILED" if bad else "all round trips ok")
sys.exit(1 if bad else 0)
