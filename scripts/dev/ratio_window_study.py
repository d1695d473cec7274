"""CPU study (reference libzstd, oracle/_ref): how much of level 3's ratio on long-range inputs comes
from the window and from the table sizes.  Development only."""
import ctypes as C, glob, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from zl_testlib import Ref

def corpus(n):
    out = bytearray()
    for f in sorted(glob.glob("/usr/lib/python3.10/*.py")):
        out += open(f, "rb").read()
        if len(out) >= n: break
    return bytes(out[:n])

def records(n, seed=1234):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    parts = []
    for i in range((n + (1 << 20) - 1) >> 20):
        b = base.copy()
        if i:
            idx = rng.integers(0, 1 << 20, size=(1 << 20) // 100)
            b[idx] = rng.integers(0, 256, size=idx.size, dtype=np.uint8)
        parts.append(b)
    return np.concatenate(parts).tobytes()[:n]

ref = Ref(); L = ref.L
def comp(data, level=3, **params):
    ids = dict(wlog=101, hlog=102, clog=103, slog=104, mml=105, tlen=106, strat=107)
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, level)
    for k, v in params.items():
        r = L.ZSTD_CCtx_setParameter(cctx, ids[k], v); assert not L.ZSTD_isError(r), (k, v)
    cap = L.ZSTD_compressBound(len(data)); dst = C.create_string_buffer(cap)
    r = L.ZSTD_compress2(cctx, dst, cap, data, len(data)); assert not L.ZSTD_isError(r)
    L.ZSTD_freeCCtx(cctx)
    return r

def chunked(data, chunk=128 << 10, **kw):
    return sum(comp(data[i:i + chunk], **kw) for i in range(0, len(data), chunk))

for name, data in (("corpus", corpus(16 << 20)), ("records", records(8 << 20))):
    n = len(data)
    print(name, n)
    print("  ref L3 whole           %.3f" % (n / comp(data)))
    for w in (16, 17, 18, 19, 20):
        print("  ref L3 wlog=%d        %.3f" % (w, n / comp(data, wlog=w)))
    print("  ref L3 chunks 128K     %.3f" % (n / chunked(data)))
    print("  ref L3 chunks 128K h13 c12 %.3f" % (n / chunked(data, hlog=13, clog=12)))
    print("  ref L3 whole hlog13 clog12 %.3f" % (n / comp(data, hlog=13, clog=12)))
    print("  ref L3 whole w17 h13 c12 %.3f" % (n / comp(data, wlog=17, hlog=13, clog=12)))
    print("  ref L1 whole           %.3f" % (n / comp(data, level=1)))
    print("  ref L5 whole           %.3f" % (n / comp(data, level=5)))
    print("  ref L7 whole           %.3f" % (n / comp(data, level=7)))
