"""Fuzz the streaming DEcoder with damaged input: reference-made and our own frames (100 bytes to 3 MB, with and without
checksum), a few flipped bits, fed to zl_decompressStream in pieces of random sizes; every other round with the route that
decodes a large frame while it arrives (round 4) opened for frames of 1 MiB.  The session must end the way the oracle
ends on the whole buffer: an error (at whatever piece) when the oracle rejects, the oracle's bytes when it accepts -- and never
hang.  usage: fuzz_stream_decode.py [rounds=100] [seed=1]"""
import sys
import time
import warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import stream
from zl_testlib import GENERATORS, Oracle, Ref, is_error
from zstdlite_amd import _lib
L = _lib.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref, o = Ref(), Oracle()
warnings.simplefilter("ignore")
pool = {k: GENERATORS[k](4 << 20, seed=81) for k in ("text", "iid10", "int20", "rand")}
base = []
for kind in pool:
    for n in (100, 5000, 131072, 400000, 3000000):
        x = pool[kind][:n]
        for cs in (False, True):
            base.append((ref.compress(x, level=int(rng.choice([1, 3, 5])), checksum=cs), n))
            base.append((z.zstd_compress(x, include_checksum=cs), n))
print(len(base), "base frames", flush=True)
bad = acc = rej = 0
t0 = time.time()
for rd in range(rounds):
    for f, n in base:
        b = bytearray(f)
        for _ in range(int(rng.integers(0, 4))):
            b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
        b = bytes(b)
        code, want = o.decompress_code(b, n + 16)
        got, err = [], None
        try:
            dctx = z.zstd_dctx()  # (held: the session uses its pointer)
            if rd & 1:  # every other round: frames of 1 MiB and more are decoded WHILE THEY ARRIVE (ZL_d_slabMiB = 1: pieces and runs of a few hundred KiB)
                L.zl_DCtx_setParameter(dctx._ptr, 9010, 1); L.zl_DCtx_setParameter(dctx._ptr, 9009, int(rng.choice([0, 16, 40])))
            dec = stream.Decompressor(dctx._ptr)
            pos = 0
            while pos < len(b):
                k = int(rng.choice([1, 64, 4096, 131702, 1 << 20]))
                got.append(dec.feed(b[pos:pos + k])); pos += k
            if not dec.at_frame_boundary:
                err = "stream ended inside a frame"
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:120]
        if is_error(code):
            rej += 1
            if err is None and b"".join(got) is not None:
                # (accepting what the oracle rejects is a defect only if the stream decoder claims a complete frame)
                bad += 1
                print("WRONG ACCEPT round", rd, "n", n, len(b), flush=True)
        else:
            acc += 1
            if err is not None or b"".join(got) != want:
                bad += 1
                print("WRONG REJECT / bytes round", rd, "n", n, err, flush=True)
print(f"{rounds} rounds, {acc} accepted, {rej} rejected, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
