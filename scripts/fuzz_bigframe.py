"""Fuzz the large-frame pipeline against the oracle: reference-made frames of 9..20 blocks get a few
damaged bytes (or are cut short); batches of them are decoded in one call.  Rules as in
fuzz_decode.py: never a hang or crash, accepted frames equal the oracle's bytes, nothing the oracle
accepts is rejected.  With a third argument "dict": every frame starts from a dictionary (trained or raw content; the reference's
frames then ride on its tables in their first block or reach into its content), the contexts and the oracle are given it.
usage: fuzz_bigframe.py [rounds] [seed] [dict]"""
import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Oracle, Ref, is_error
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o = Oracle(); L = _lib.lib(); ref = Ref()
use_dict = len(sys.argv) > 3 and sys.argv[3] == "dict"
D = None
if use_dict:
    t_ = GENERATORS["text"](2 << 20, seed=99)
    D = ref.train_dict([t_[i * 3000:(i + 1) * 3000] for i in range(400)], 20000) if rng.random() < 0.6 else t_[70000:110000]
    print("dictionary:", "trained" if D[:4] == bytes([0x37, 0xA4, 0x30, 0xEC]) else "raw content", len(D), "bytes")
dkw = {"dict": D} if use_dict else {}
base = []
for gen, n, kw in (("text", 1200000, {"level": 3}), ("iid10", 1300000, {"level": 3, "checksum": True}), ("int20", 1500000, {"level": 1}),
                   ("text", 2500000, {"level": 9}), ("iid10", 1100000, {"level": 19})):
    d = GENERATORS[gen](n, seed=n) + (bytes(200000) if gen == "int20" else b"")
    base.append((ref.compress(d, **kw, **dkw), len(d)))
# frames of this library's own one-frame layout: segments of several blocks that share tables, far matches across chunks
from zl_testlib import LONG_RANGE
for d in (GENERATORS["text"](900000, seed=5) + GENERATORS["iid10"](500000, seed=6), LONG_RANGE["records"](3 << 20)[: (2 << 20) + 300000], LONG_RANGE["corpus"](1500000)):
    base.append((z.zstd_compress(d, **dkw), len(d)))
print(len(base), "base frames;", sum(len(f) for f, _ in base), "compressed bytes")
dctx = z.zstd_dctx(**dkw)
bad_accept = bad_reject = accepted = rejected = 0
t0 = time.time()
for rd in range(rounds):
    frames = []
    for f, n in base:
        for k in range(6):
            b = bytearray(f)
            if k == 5:
                b = b[: int(rng.integers(8, len(b)))]
            else:
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
        code, out = o.decompress_code(f, n, **dkw)
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
    # the same damaged frames through the host entry, one by one: the host walks the blocks and uploads what it saw
    for i, (f, n) in enumerate(frames[:: 3]):
        code, out = o.decompress_code(f, n, **dkw)
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        r1 = L.zl_decompressDCtx(dctx._ptr, buf.ctypes.data, n, f, len(f))
        if not is_error(r1):
            accepted += 1
            if is_error(code) or r1 != code or buf[:r1].tobytes() != out:
                bad_accept += 1
                print("host entry accepted, oracle disagrees: frame", 3 * i, "gpu", r1, "oracle", code if not is_error(code) else o.error_name(code))
        else:
            rejected += 1
            if not is_error(code):
                bad_reject += 1
                print("host entry rejected a frame the oracle accepts: frame", 3 * i, _lib.error_name(r1), "oracle size", code)
    print(f"round {rd}: {accepted} accepted, {rejected} rejected, {bad_accept} wrong accepts, {bad_reject} wrong rejects, large frames so far {L.zl_dctx_large_frames(dctx._ptr)}, {time.time() - t0:.0f}s", flush=True)
assert bad_accept == 0 and bad_reject == 0
print("ok")
