"""Fuzz the GPU decoder against the oracle: frames from the HIP encoder (multi-block, shared tables) and
from the reference library get a few damaged bytes; every frame of a batch is decoded in one launch.
Rules: the GPU never hangs or crashes; when it accepts a frame its bytes equal the oracle's, and it
never rejects a frame the oracle accepts (the GPU decoders are as strict as the oracle)."""
import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Oracle, Ref, have_ref, is_error
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
single = len(sys.argv) > 3 and sys.argv[3] in ("single", "wide")  # only single-block reference frames: the fused entropy kernel with its quad chains
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"  # ... in batches of more than 1024 frames: the two-part schedule of zl2_launch
o = Oracle(); L = _lib.lib()
base = []
for gen in ("text", "iid10", "int20"):
    for n in (300, 5000, 20000, 70000, 131072):
        d = GENERATORS[gen](n, seed=n)
        if not single: base.append((z.zstd_compress(d), n))
if have_ref():
    ref = Ref()
    for gen in ("text", "iid10"):
        for n in ((5000, 30000, 65536, 100000, 131072) if single else (5000, 65536, 131072, 300000)):
            d = GENERATORS[gen](n, seed=n + 1)
            base.append((ref.compress(d, level=3), n))
print(len(base), "base frames;", sum(len(f) for f, _ in base), "compressed bytes")
dctx = z.zstd_dctx()
bad_accept = bad_reject = accepted = rejected = 0
t0 = time.time()
for rd in range(rounds):
    frames = []
    for f, n in base:
        for _ in range(110 if wide else 6):
            b = bytearray(f)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(4, len(b)))  # keep the magic number
                b[pos] ^= 1 << int(rng.integers(0, 8))
            frames.append((bytes(b), n))
    m = len(frames)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for _, n in frames]
    srcs = [np.frombuffer(f, dtype=np.uint8) for f, _ in frames]
    dsts = (C.c_void_p * m)(*[x.ctypes.data for x in outs])
    caps = (C.c_size_t * m)(*[n for _, n in frames])
    ptrs = (C.c_void_p * m)(*[x.ctypes.data for x in srcs])
    sizes = (C.c_size_t * m)(*[len(f) for f, _ in frames])
    res = (C.c_size_t * m)()
    r = L.zl_decompress_batch(dctx._ptr, m, dsts, caps, ptrs, sizes, res)
    assert not is_error(r), _lib.error_name(r)
    for i, (f, n) in enumerate(frames):
        code, out = o.decompress_code(f, n)
        g_ok = not is_error(res[i])
        if g_ok:
            accepted += 1
            if is_error(code) or res[i] != code or outs[i][: res[i]].tobytes() != out:
                bad_accept += 1
                print("GPU accepted, oracle disagrees: frame", i, "gpu", res[i], "oracle", code if not is_error(code) else o.error_name(code))
        else:
            rejected += 1
            if not is_error(code):
                bad_reject += 1
                print("GPU rejected a frame the oracle accepts: frame", i, _lib.error_name(res[i]), "oracle size", code)
print(f"{rounds} rounds, {accepted} accepted, {rejected} rejected, wrong accepts {bad_accept}, wrong rejects {bad_reject}, {time.time() - t0:.1f}s")
sys.exit(1 if bad_accept or bad_reject else 0)
