#!/usr/bin/env python3
"""Dev: levels 3 - 5 on small and middling inputs: size against the reference's at the same level and the host call's wall time.
Run once per library (ZL_HIP_LIB=... for the variant built with -DZL_DEEP_SMALL_MAX=0: the table parse at every size)."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
corpus = LONG_RANGE["corpus"](None)
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 4, 5]
bad = 0
for kind in ("corpus", "text", "records", "int20"):
    for n in (20 << 10, 64 << 10, 200_000, 1 << 20, 4 << 20, 11_000_000, 16 << 20, 32 << 20, 64 << 20):
        if kind == "corpus":
            if n > len(corpus): continue
            data = corpus[len(corpus) // 3:][:n] if n < len(corpus) // 2 else corpus[:n]
        elif kind == "records": data = LONG_RANGE["records"](n)
        else: data = GENERATORS[kind](n)
        for lvl in levels:
            cctx = z.zstd_cctx(level=lvl)
            best = 1e9
            for _ in range(5):
                t0 = time.time(); comp = z.zstd_compress(data, cctx=cctx); best = min(best, time.time() - t0)
            ok = ref.decompress(comp, len(data)) == data and z.zstd_decompress(comp) == data
            bad += not ok
            theirs = len(ref.compress(data, level=lvl))
            print(f"{kind:8s} {n:9d} level {lvl} size/ref {len(comp)/theirs:.3f} {'ok' if ok else 'MISMATCH'} {best*1e3:8.3f} ms", flush=True)
print("FAILED" if bad else "all round trips ok")
sys.exit(1 if bad else 0)
