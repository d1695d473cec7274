"""Fuzz the host entries on LARGE inputs (60 MiB to 700 MiB: several sections, the pipelined and the sharded paths): random
layout, checksum, seek table, level and number of device slots; the bytes must not depend on the slots, the reference must
decode them to the input, and so must our decoder (with its own random slot count).  usage: fuzz_big.py [rounds=12] [seed=1]"""
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref = Ref()
L = _lib.lib()
pool = {k: np.frombuffer(GENERATORS[k](8 << 20, seed=41), dtype=np.uint8) for k in ("text", "iid10", "int20", "rand")}


def make(n):
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    while pos < n:
        k = int(min(n - pos, int(rng.integers(1, 6 << 20))))
        r = rng.random()
        if r < 0.8 or pos == 0:
            src = pool[str(rng.choice(list(pool)))]
            o = int(rng.integers(0, len(src) - k)) if k < len(src) else 0
            out[pos:pos + k] = src[o:o + k]
        elif r < 0.9:
            out[pos:pos + k] = int(rng.integers(0, 256))                       # a run
        else:
            o = int(rng.integers(0, pos))                                       # a copy of something earlier, any distance
            k = min(k, pos - o)
            out[pos:pos + k] = out[o:o + k]
        pos += k
    return out


def compress(host, layout, cs, seek, level, slots):
    cctx = L.zl_createCCtx()
    L.zl_CCtx_setParameter(cctx, 9003, layout); L.zl_CCtx_setParameter(cctx, 201, cs); L.zl_CCtx_setParameter(cctx, 9006, seek)
    L.zl_CCtx_setParameter(cctx, 100, level); L.zl_CCtx_setParameter(cctx, 9007, slots)
    cap = L.zl_compressBound(host.size)
    buf = np.empty(cap, dtype=np.uint8)
    r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, host.size)
    L.zl_freeCCtx(cctx)
    assert not _lib.is_error(r), _lib.error_name(r)
    return buf[:r]


bad = 0
t0 = time.time()
for rd in range(rounds):
    n = int(rng.integers(60 << 20, 700 << 20))
    layout, cs, seek, level = int(rng.choice([0, 1, 2])), int(rng.random() < 0.3), int(rng.random() < 0.5), int(rng.choice([3, 3, 5, 6, 9, 12, 17, 21]))
    s1, s2, s3 = 1, int(rng.choice([2, 3])), int(rng.choice([1, 2, 3]))
    print("round", rd, "MiB", n >> 20, dict(layout=layout, checksum=cs, seek=seek, level=level, slots=(s1, s2, s3)), flush=True)
    try:
        host = make(n)
        a = compress(host, layout, cs, seek, level, s1)
        b = compress(host, layout, cs, seek, level, s2)
        assert a.size == b.size and np.array_equal(a, b), "the bytes depend on the number of device slots"
        back = ref.decompress(a.tobytes(), n)
        assert back == host.tobytes(), "reference decoder disagrees"
        del back
        dctx = L.zl_createDCtx()
        L.zl_DCtx_setParameter(dctx, 9008, s3)
        out = np.zeros(n, dtype=np.uint8)
        r2 = L.zl_decompressDCtx(dctx, out.ctypes.data, n, a.ctypes.data, a.size)
        L.zl_freeDCtx(dctx)
        assert r2 == n and np.array_equal(out, host), ("our decoder disagrees", _lib.error_name(r2) if _lib.is_error(r2) else r2)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL round", rd, repr(e)[:300], flush=True)
print(f"{rounds} rounds, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
