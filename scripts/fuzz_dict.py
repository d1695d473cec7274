"""Fuzz the decoders on frames that start from a dictionary (lane pipeline with inherited tables, general kernel for what it hands
back) against the oracle: reference-made frames of 100 bytes to 600 KB, trained and raw-content dictionaries, with and without
checksum, a few flipped bits each; every frame of a round is decoded in one launch.  Rules as scripts/fuzz_decode.py: no hang,
no crash; accepted exactly when the oracle accepts, and then the same bytes.  usage: fuzz_dict.py [rounds=20] [seed=1]"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Oracle, Ref, is_error
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o, ref, L = Oracle(), Ref(), _lib.lib()
text = GENERATORS["text"](4 << 20, seed=77)
dicts = [ref.train_dict([text[i * 3000:(i + 1) * 3000] for i in range(400)], 12000), text[50000:90000]]
base = []
for di, d in enumerate(dicts):
    for n in (100, 700, 2048, 5000, 30000, 131072, 200000, 600000):
        x = text[(2 << 20) + 3 * n:(2 << 20) + 4 * n]
        for cs in (False, True):
            base.append((di, ref.compress(x, level=3, checksum=cs, dict=d), n))
    x = GENERATORS["iid10"](40000, seed=5)
    base.append((di, ref.compress(x, level=3, dict=d), len(x)))
print(len(base), "base frames")
bad_accept = bad_reject = accepted = rejected = 0
t0 = time.time()
for rnd in range(rounds):
    for di, d in enumerate(dicts):
        frames = []
        for dj, f, n in base:
            if dj != di:
                continue
            for _ in range(12):
                b = bytearray(f)
                for _ in range(int(rng.integers(0, 4))):
                    b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
                frames.append((bytes(b), n))
        m = len(frames)
        outs = [np.zeros(max(n, 1), dtype=np.uint8) for _, n in frames]
        srcs = [np.frombuffer(f, dtype=np.uint8) for f, _ in frames]
        res = (C.c_size_t * m)()
        dctx = z.zstd_dctx(dict=d)
        r = L.zl_decompress_batch(dctx._ptr, m, (C.c_void_p * m)(*[x.ctypes.data for x in outs]), (C.c_size_t * m)(*[n for _, n in frames]),
                                  (C.c_void_p * m)(*[x.ctypes.data for x in srcs]), (C.c_size_t * m)(*[len(f) for f, _ in frames]), res)
        assert not is_error(r)
        for i, (f, n) in enumerate(frames):
            code, out = o.decompress_code(f, n, dict=d)
            if is_error(res[i]):
                rejected += 1
                if not is_error(code):
                    bad_reject += 1
                    print("WRONG REJECT", rnd, di, i, _lib.error_name(res[i]))
            else:
                accepted += 1
                if is_error(code) or res[i] != code or outs[i][: res[i]].tobytes() != out:
                    bad_accept += 1
                    print("WRONG ACCEPT", rnd, di, i)
print(f"{rounds} rounds, {accepted} accepted, {rejected} rejected, wrong accepts {bad_accept}, wrong rejects {bad_reject}, {time.time() - t0:.1f}s")
sys.exit(1 if bad_accept or bad_reject else 0)
