"""Fuzz the entries that take MANY frames or objects: (1) zl_compress_batch / zl_decompress_batch on random lists (empty objects,
objects of several chunks, with and without a dictionary) against the one-object entries and the reference decoder; (2)
zl_decompressDCtx on buffers of several frames -- ours in both layouts, the reference's, skippable frames, seek tables in between
-- against the reference's ZSTD_decompress of the same buffer, whole and cut short at a random place (an error, never a crash;
zstd.c:41722-41760).  usage: fuzz_multi.py [rounds=100] [seed=1]"""
import struct
import sys
import time
import warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zl_testlib import GENERATORS, Ref, is_error
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref = Ref()
warnings.simplefilter("ignore")
pool = {k: GENERATORS[k](2 << 20, seed=31) for k in ("text", "iid10", "int20", "rand")}
d_trained = ref.train_dict([pool["text"][i * 1500:(i + 1) * 1500] for i in range(400)], 6000)


def obj(maxn):
    n = int(rng.choice([0, 1, 5, 64, 300, 4000, int(rng.integers(0, maxn))]))
    src = pool[str(rng.choice(list(pool)))]
    o = int(rng.integers(0, len(src) - n)) if n < len(src) else 0
    return src[o:o + n]


bad = 0
t0 = time.time()
for rd in range(rounds):
    try:
        # ---- (1) batches
        items = [obj(int(rng.choice([2000, 2000, 200000, 600000]))) for _ in range(int(rng.integers(1, 400)))]
        use_d = rng.random() < 0.4
        kw = {"dict": d_trained} if use_d else {}
        cctx = z.zstd_cctx(**kw)
        comps = z.zstd_compress_batch(items, cctx=cctx)
        k = int(rng.integers(0, len(items)))
        # (one object of more than 16 KiB without a dictionary is searched by the deep match finder, the objects of a batch by the tables:
        # both valid, not the same bytes -- DESIGN.md 3.3)
        one = z.zstd_compress(items[k], cctx=cctx, frame_layout="frames")
        if not (comps[k] == one or len(items[k]) > 131072 or (len(items[k]) > 16384 and not use_d)):
            import os, pickle
            os.makedirs("gpurun_out", exist_ok=True)
            with open(f"gpurun_out/multi_fail_{rd}.pkl", "wb") as fh:
                pickle.dump({"item": items[k], "batch": comps[k], "one": one, "dict": d_trained if use_d else None, "items": items, "k": k}, fh)
        assert comps[k] == one or len(items[k]) > 131072 or (len(items[k]) > 16384 and not use_d), ("batch item differs from the one-object entry", len(items), use_d, len(items[k]), len(comps[k]), len(one), comps[k][:16].hex(), one[:16].hex())
        assert z.zstd_decompress(one, **kw) == items[k] or len(items[k]) > 131072
        for i in rng.integers(0, len(items), 5):
            assert ref.decompress(comps[int(i)], max(len(items[int(i)]), 1), dict=d_trained if use_d else None) == items[int(i)], "reference decoder disagrees on a batch item"
        small = [(c, x) for c, x in zip(comps, items) if len(x) <= 131072]
        if small:
            assert z.zstd_decompress_batch([c for c, _ in small], **kw) == [x for _, x in small], "batch decode differs"
        # ---- (2) a buffer of several frames
        parts, want = [], []
        for _ in range(int(rng.integers(1, 7))):
            x = obj(int(rng.choice([3000, 300000, 1500000])))
            r = rng.random()
            if r < 0.35:
                f = z.zstd_compress(x, include_checksum=bool(rng.random() < 0.3), seek_table=bool(rng.random() < 0.3))
            elif r < 0.6:
                f = z.zstd_compress(x, frame_layout="frames", seek_table=bool(rng.random() < 0.3))
            else:
                f = ref.compress(x, level=int(rng.choice([1, 3, 6])), checksum=bool(rng.random() < 0.3))
            parts.append(f); want.append(x)
            if rng.random() < 0.3:
                pay = bytes(rng.integers(0, 256, int(rng.integers(0, 50)), dtype=np.uint8))
                parts.append(struct.pack("<II", 0x184D2A50 + int(rng.integers(0, 16)), len(pay)) + pay)
        buf, plain = b"".join(parts), b"".join(want)
        code, out = ref.decompress_code(buf, max(len(plain), 1))
        assert not is_error(code) and out == plain, "the reference does not read this buffer (fuzzer bug?)"
        L = z._lib.lib()
        back = np.zeros(max(len(plain), 1), dtype=np.uint8)
        dctx = z.zstd_dctx()
        r2 = L.zl_decompressDCtx(dctx._ptr, back.ctypes.data, len(plain), buf, len(buf))
        assert r2 == len(plain) and back[:len(plain)].tobytes() == plain, ("multi-frame decode differs", z._lib.error_name(r2) if is_error(r2) else r2)
        cut = int(rng.integers(0, len(buf)))
        rc, _ = ref.decompress_code(buf[:cut], max(len(plain), 1))
        r3 = L.zl_decompressDCtx(dctx._ptr, back.ctypes.data, len(plain), buf[:cut], cut)
        assert is_error(r3) == is_error(rc), ("a buffer cut short", cut, len(buf), z._lib.error_name(r3) if is_error(r3) else r3, rc if not is_error(rc) else "ref error")
        if not is_error(rc):
            assert r3 == rc
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL round", rd, repr(e)[:400], flush=True)
print(f"{rounds} rounds, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
