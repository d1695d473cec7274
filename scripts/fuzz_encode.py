"""Fuzz the encoder: random inputs (mixtures of the generators, runs, copies from earlier in the input, lengths from 0 to a few
chunks with the edges of chunks and blocks over-represented), random options (level, checksum, dictionary, layout, seek table):
every output must be ONE or more standard frames that the REFERENCE decoder turns back into the input (and our decoder too),
and the same call must write the same bytes twice.  usage: fuzz_encode.py [rounds=200] [seed=1] [largest input=400000]"""
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zl_testlib import GENERATORS, Ref
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
maxn = int(sys.argv[3]) if len(sys.argv) > 3 else 400000  # (largest input; a few MiB exercise many chunks per frame)
ref = Ref()
kinds = ["text", "iid10", "int20", "rand"]
pool = {k: GENERATORS[k](1 << 20, seed=9) for k in kinds}
dict_trained = ref.train_dict([pool["text"][i * 2000:(i + 1) * 2000] for i in range(300)], 8000)
dict_raw = pool["text"][300000:320000]
edges = [0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 255, 256, 257, 1023, 16383, 16384, 16385, 65535, 65536, 131071, 131072, 131073, 262143, 262144, 262145, 393217]


def make():
    if rng.random() < 0.4:
        n = int(edges[int(rng.integers(0, len(edges)))])
    else:
        n = int(rng.integers(0, maxn))
    parts, have = [], 0
    while have < n:
        r = rng.random()
        k = int(min(n - have, max(1, int(rng.integers(1, max(60000, maxn // 20)))), (1 << 20) - 1))  # (a pool holds 1 MiB)
        if r < 0.55:
            src = pool[kinds[int(rng.integers(0, 4))]]
            o = int(rng.integers(0, len(src) - k))
            p = src[o:o + k]
        elif r < 0.7:
            p = bytes([int(rng.integers(0, 256))]) * k                      # a run
        elif r < 0.9 and have:
            whole = b"".join(parts)
            o = int(rng.integers(0, have))
            p = (whole[o:o + k] * (k // max(1, len(whole[o:o + k])) + 1))[:k]  # a copy of something earlier
        else:
            p = bytes(rng.integers(0, 4, k, dtype=np.uint8))                # four symbols
        parts.append(p)
        have += len(p)
    return b"".join(parts)[:n]


bad = 0
t0 = time.time()
for rd in range(rounds):
    x = make()
    opts = {"level": int(rng.choice([1, 2, 3, 3, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, -5, -1, 0, 4, 16, 17, 19, 19, 20, 22]))}  # (src/cctx.c:261-268: -5 .. 22; every level is an operating point: 16 ... 22 the price-based parse)
    if rng.random() < 0.3:
        opts["include_checksum"] = True
    d = None
    if rng.random() < 0.3:
        d = dict_trained if rng.random() < 0.6 else dict_raw
        opts["dict"] = d
    if rng.random() < 0.5:
        opts["frame_layout"] = str(rng.choice(["frames", "one_frame"]))
    if rng.random() < 0.2:
        opts["seek_table"] = True
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c1 = z.zstd_compress(x, **opts)
            c2 = z.zstd_compress(x, **opts)
        assert c1 == c2, "not deterministic"
        back = ref.decompress(c1, max(len(x), 1), dict=d)
        assert back == x, "reference decoder disagrees"
        dopts = {"dict": d} if d is not None else {}
        if len(x) and not (opts.get("frame_layout") == "frames" and len(x) > 131072):
            assert z.zstd_decompress(c1, **dopts) == x, "our decoder disagrees"
    except Exception as e:  # noqa: BLE001
        bad += 1
        try:  # (the failing input, for scripts/dev work)
            import json, os
            os.makedirs("gpurun_out", exist_ok=True)
            open(f"gpurun_out/fuzz_encode_fail_{rd}.bin", "wb").write(x)
            json.dump({k: v for k, v in opts.items() if k != "dict"} | {"dict": None if d is None else ("trained" if d is dict_trained else "raw")}, open(f"gpurun_out/fuzz_encode_fail_{rd}.json", "w"))
        except Exception:
            pass
        print("FAIL round", rd, "n", len(x), opts.keys(), {k: v for k, v in opts.items() if k != "dict"}, repr(e)[:200], flush=True)
print(f"{rounds} rounds, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
