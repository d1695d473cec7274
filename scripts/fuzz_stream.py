"""Fuzz the streaming entries (zl_compressStream2 / zl_decompressStream behind zstdlite_amd.stream): an input is fed in pieces of
random sizes with flushes at random places, with or without a pledged size, with and without checksum; what comes out must be
standard frames that the REFERENCE turns back into the input, a pledged session without flushes must write the one-shot entry's
bytes, and our streaming decoder fed the result (or a reference-made stream) in random pieces must return the input.
usage: fuzz_stream.py [rounds=60] [seed=1]"""
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import stream
from zl_testlib import GENERATORS, Ref
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref = Ref()
verbose = len(sys.argv) > 3
pool = {k: GENERATORS[k](4 << 20, seed=21) for k in ("text", "iid10", "int20", "rand")}


def make():
    n = int(rng.choice([0, 1, 100, 131072, 131073, 500000, int(rng.integers(0, 3 << 20)), int(rng.integers(0, 300000))]))
    if rng.random() < 0.02:
        n = int(rng.integers(60 << 20, 150 << 20))  # (above the 64 MiB job of the streaming compressor: several jobs, one frame when pledged)
    parts, have = [], 0
    while have < n:
        k = int(min(n - have, int(rng.integers(1, 400000))))
        src = pool[str(rng.choice(list(pool)))]
        o = int(rng.integers(0, len(src) - k))
        parts.append(src[o:o + k]); have += k
    return b"".join(parts)


def pieces(x):
    out, pos = [], 0
    while pos < len(x):
        k = int(rng.choice([1, 7, 4096, 131072, 131702, 1 << 20, int(rng.integers(1, 2 << 20))]))
        out.append(x[pos:pos + k]); pos += k
    return out


bad = 0
t0 = time.time()
import json
import os
if os.environ.get("FUZZ_STREAM_STATE"):  # (replay: the generator state a crashed run left in gpurun_out/fuzz_stream_state.json)
    rng.bit_generator.state = json.load(open(os.environ["FUZZ_STREAM_STATE"]))
for rd in range(rounds):
    if verbose:
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(rng.bit_generator.state, open("gpurun_out/fuzz_stream_state.json", "w"))
    x = make()
    checksum = bool(rng.random() < 0.3)
    pledged = bool(rng.random() < 0.5)
    flushes = bool(rng.random() < 0.4)
    if verbose:
        print("round", rd, "n", len(x), dict(checksum=checksum, pledged=pledged, flushes=flushes), flush=True)
    try:
        cctx = z.zstd_cctx(include_checksum=checksum)
        comp = stream.Compressor(cctx._ptr, pledged=len(x) if pledged else None)
        out = []
        ps = pieces(x)
        for i, p in enumerate(ps):
            out.append(comp.feed(p))
            if flushes and rng.random() < 0.3:
                out.append(comp.flush())
        out.append(comp.finish())
        c = b"".join(out)
        if verbose:
            print("   stream compress done", len(c), "pieces", len(ps), "smallest", min(map(len, ps), default=0), flush=True)
        assert ref.decompress(c, max(len(x), 1)) == x, "reference decoder disagrees"
        if pledged and not flushes:
            one = z.zstd_compress(x, include_checksum=checksum)
            if verbose:
                print("   one-shot done", len(one), flush=True)
            assert c == one, "a pledged session is not the one-shot call"
        for si, source in enumerate((c, ref.compress(x, level=int(rng.choice([1, 3, 7])), checksum=checksum))):
            if verbose:
                print("   decode source", si, len(source), flush=True)
            dctx = z.zstd_dctx()
            dec = stream.Decompressor(dctx._ptr)
            got = b"".join(dec.feed(p) for p in pieces(source))
            assert got == x, "streaming decoder disagrees"
            assert dec.at_frame_boundary or len(source) == 0
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL round", rd, "n", len(x), dict(checksum=checksum, pledged=pledged, flushes=flushes), repr(e)[:300], flush=True)
print(f"{rounds} rounds, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
