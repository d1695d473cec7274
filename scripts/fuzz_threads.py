"""Several host threads at once, each with contexts of its own (the reference's glue is re-entrant per context: src/cctx.c:191):
one-shot calls, batches of small objects (whose host loops share the library's thread pool), streams and large one-frame
inputs, side by side for a while; every result is checked against the input and, for a sample, the reference decoder.
usage: fuzz_threads.py [seconds=20] [threads=6] [seed=1]"""
import sys
import threading
import time
import warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import stream
from zl_testlib import GENERATORS, Ref
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
warnings.simplefilter("ignore")
pool = {k: GENERATORS[k](16 << 20, seed=61) for k in ("text", "iid10", "int20", "rand")}
errors, counts = [], [0] * nthreads
stop = time.time() + seconds


def worker(w):
    rng = np.random.default_rng(seed * 100 + w)
    ref = Ref()
    try:
        while time.time() < stop:
            kind = int(rng.integers(0, 4))
            src = pool[str(rng.choice(list(pool)))]
            if kind == 0:    # a batch of small objects
                items = [src[int(o):int(o) + int(k)] for o, k in zip(rng.integers(0, len(src) - 3000, 20000), rng.integers(0, 3000, 20000))]
                comps = z.zstd_compress_batch(items)
                assert z.zstd_decompress_batch(comps) == items
                assert ref.decompress(comps[7], max(len(items[7]), 1)) == items[7]
            elif kind == 1:  # one-shot, medium
                n = int(rng.integers(1, 12 << 20)); o = int(rng.integers(0, len(src) - n))
                x = src[o:o + n]
                c = z.zstd_compress(x, include_checksum=bool(rng.random() < 0.5))
                assert z.zstd_decompress(c) == x
            elif kind == 2:  # a stream
                n = int(rng.integers(1, 8 << 20)); o = int(rng.integers(0, len(src) - n))
                x = src[o:o + n]
                cctx = z.zstd_cctx()
                comp = stream.Compressor(cctx._ptr, pledged=n)
                out = [comp.feed(x[p:p + 300000]) for p in range(0, n, 300000)]
                out.append(comp.finish())
                assert ref.decompress(b"".join(out), n) == x
            else:            # one large frame through the large-frame decoder
                x = src * int(rng.integers(1, 5))
                c = ref.compress(x, level=1)
                assert z.zstd_decompress(c) == x
            counts[w] += 1
    except Exception as e:  # noqa: BLE001
        errors.append((w, repr(e)[:300]))


th = [threading.Thread(target=worker, args=(w,)) for w in range(nthreads)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"{nthreads} threads, {seconds:.0f} s: operations per thread {counts}, errors {errors}")
sys.exit(1 if errors else 0)
