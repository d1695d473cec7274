#!/usr/bin/env python3
"""Dev: what the reference's levels owe to window, search depth and strategy: the corpus through the
reference library with single parameters overridden.  CPU only."""
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
L = ref.L
P = dict(level=100, wlog=101, hlog=102, clog=103, slog=104, mml=105, tlen=106, strat=107)
def comp(data, **kw):
    cctx = L.ZSTD_createCCtx()
    for k, v in kw.items():
        r = L.ZSTD_CCtx_setParameter(cctx, P[k], v); assert not L.ZSTD_isError(r), (k, v)
    cap = L.ZSTD_compressBound(len(data)); dst = C.create_string_buffer(cap)
    r = L.ZSTD_compress2(cctx, dst, cap, data, len(data)); L.ZSTD_freeCCtx(cctx)
    return r
name = sys.argv[1] if len(sys.argv) > 1 else "corpus"
data = LONG_RANGE[name](None) if name in LONG_RANGE else GENERATORS[name](16 << 20)
n = len(data)
for lvl in (3, 5, 6, 7, 8, 9, 10, 12):
    row = [f"L{lvl:2d} full {n / comp(data, level=lvl):.3f}"]
    for w in (16, 17, 18, 19, 20):
        row.append(f"w{w} {n / comp(data, level=lvl, wlog=w):.3f}")
    print("  ".join(row), flush=True)
