"""Fuzz context STATE through the C-ABI: ONE compression and ONE decompression context live for the whole run while parameters
change between calls -- level, checksum, layout, seek table; a trained dictionary, a raw-content one, none (loadDictionary with
NULL) -- and calls of every kind follow each other (one-shot, batch, stream session, reset).  After every call: the reference
decodes what came out with the dictionary that was loaded at the time, our decoder does too, and a context that has just had its
dictionary changed or removed must not still use the old one (its digested tables, its prebuilt lane tables, its hash).
usage: fuzz_ctx.py [rounds=300] [seed=1]"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref, is_error
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref = Ref()
L = _lib.lib()
text = GENERATORS["text"](6 << 20, seed=91)
other = GENERATORS["iid10"](6 << 20, seed=92)
dicts = {"trained_a": ref.train_dict([text[i * 2000:(i + 1) * 2000] for i in range(300)], 9000),
         "trained_b": ref.train_dict([other[i * 2000:(i + 1) * 2000] for i in range(300)], 5000),
         "raw": text[400000:430000], None: None}
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
cur_c = cur_d = None
bad = 0
t0 = time.time()


def set_dict(ctx, is_c, name):
    d = dicts[name]
    fn = L.zl_CCtx_loadDictionary if is_c else L.zl_DCtx_loadDictionary
    r = fn(ctx, d, len(d) if d else 0)
    assert not is_error(r), _lib.error_name(r)


for rd in range(rounds):
    try:
        if rng.random() < 0.35:
            cur_c = list(dicts)[int(rng.integers(0, 4))]
            set_dict(cctx, True, cur_c)
        if rng.random() < 0.2:
            L.zl_CCtx_reset(cctx, int(rng.choice([1, 2])))  # session only / parameters (zstd.h:590; parameters also drops the dictionary)
            if rng.random() < 0.0:
                pass
        level = int(rng.choice([1, 3, 5, 7, 12, -1]))
        cs = int(rng.random() < 0.4)
        layout = int(rng.choice([0, 1, 2]))
        seek = int(rng.random() < 0.2)
        # (a reset of the parameters may have dropped dictionary and parameters: set everything again, dictionary included)
        set_dict(cctx, True, cur_c)
        for prm, v in ((100, level), (201, cs), (9003, layout), (9006, seek)):
            assert not is_error(L.zl_CCtx_setParameter(cctx, prm, v))
        n = int(rng.choice([0, 50, 3000, 131072, 131073, int(rng.integers(0, 3 << 20))]))
        src = text if rng.random() < 0.7 else other
        o = int(rng.integers(0, len(src) - n))
        x = src[o:o + n]
        cap = L.zl_compressBound(n)
        buf = np.empty(max(cap, 1), dtype=np.uint8)
        r = L.zl_compress2(cctx, buf.ctypes.data, cap, x, n)
        assert not is_error(r), _lib.error_name(r)
        c = buf[:r].tobytes()
        dc = dicts[cur_c]
        assert ref.decompress(c, max(n, 1), dict=dc) == x, ("reference decoder disagrees", cur_c, level, cs, layout, seek, n)
        # the decoder context follows with a dictionary of its own choosing first: a wrong or missing one must fail or -- for a raw
        # dictionary, which frames do not name -- give whatever the reference gives with the same wrong dictionary
        if rng.random() < 0.3:
            cur_d = list(dicts)[int(rng.integers(0, 4))]
            set_dict(dctx, False, cur_d)
            rc, rout = ref.decompress_code(c, max(n, 1), dict=dicts[cur_d])
            back = np.zeros(max(n, 1), dtype=np.uint8)
            r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, c, len(c))
            if is_error(rc):
                assert is_error(r2), ("the reference rejects with this dictionary, we return", r2, cur_c, cur_d)
            else:
                assert r2 == rc and back[:r2].tobytes() == rout, ("differs from the reference with the same (possibly wrong) dictionary", cur_c, cur_d, n)
        cur_d = cur_c
        set_dict(dctx, False, cur_d)
        back = np.zeros(max(n, 1), dtype=np.uint8)
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, c, len(c))
        assert r2 == n and back[:n].tobytes() == x, ("our decoder disagrees", _lib.error_name(r2) if is_error(r2) else r2, cur_c, n)
        # a batch on the same contexts
        if rng.random() < 0.5:
            m = int(rng.integers(1, 60))
            items = [text[int(p):int(p) + int(k)] for p, k in zip(rng.integers(0, len(text) - 70000, m), rng.choice([0, 40, 900, 5000, 70000], m))]
            sizes = (C.c_size_t * m)(*[len(i) for i in items])
            caps = [L.zl_compressBound(len(i)) for i in items]
            outs = [np.empty(max(k, 1), dtype=np.uint8) for k in caps]
            res = (C.c_size_t * m)()
            srcs = [np.frombuffer(i, dtype=np.uint8) if len(i) else np.zeros(1, dtype=np.uint8) for i in items]
            rr = L.zl_compress_batch(cctx, m, (C.c_void_p * m)(*[q.ctypes.data for q in outs]), (C.c_size_t * m)(*caps), (C.c_void_p * m)(*[q.ctypes.data for q in srcs]), sizes, res)
            assert not is_error(rr)
            for i in range(m):
                assert not is_error(res[i])
                f = outs[i][:res[i]].tobytes()
                assert ref.decompress(f, max(len(items[i]), 1), dict=dc) == items[i], ("batch item", i, len(items[i]), cur_c)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL round", rd, repr(e)[:400], flush=True)
print(f"{rounds} rounds, failures {bad}, {time.time() - t0:.1f}s")
L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
sys.exit(1 if bad else 0)
