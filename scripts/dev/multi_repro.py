"""Dev: replay scripts/fuzz_multi.py's rounds (same seed) and say how a batch item differs from the one-object entry."""
import struct, sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zl_testlib import GENERATORS, Ref
seed = int(sys.argv[1]); want_rd = [int(x) for x in sys.argv[2].split(",")]
rng = np.random.default_rng(seed)
ref = Ref()
warnings.simplefilter("ignore")
pool = {k: GENERATORS[k](2 << 20, seed=31) for k in ("text", "iid10", "int20", "rand")}
d_trained = ref.train_dict([pool["text"][i * 1500:(i + 1) * 1500] for i in range(400)], 6000)
def obj(maxn):
    n = int(rng.choice([0, 1, 5, 64, 300, 4000, int(rng.integers(0, maxn))]))
    src = pool[str(rng.choice(list(pool)))]
    o = int(rng.integers(0, len(src) - n)) if n < len(src) else 0
    return src[o:o + n]
for rd in range(max(want_rd) + 1):
    items = [obj(int(rng.choice([2000, 2000, 200000, 600000]))) for _ in range(int(rng.integers(1, 400)))]
    use_d = rng.random() < 0.4
    kw = {"dict": d_trained} if use_d else {}
    k = None
    if rd in want_rd:
        cctx = z.zstd_cctx(**kw)
        comps = z.zstd_compress_batch(items, cctx=cctx)
        k = int(rng.integers(0, len(items)))
        one = z.zstd_compress(items[k], cctx=cctx, frame_layout="frames")
        again = z.zstd_compress_batch(items, cctx=cctx)
        alone = z.zstd_compress_batch([items[k]], cctx=cctx)
        print(f"round {rd}: {len(items)} items, dict {use_d}, k {k}, len {len(items[k])}, batch {len(comps[k])} one {len(one)} equal {comps[k] == one}; batch twice equal {again == comps}; batch of one equal to one {alone[0] == one}")
        print("  batch:", comps[k][:24].hex(), " one:", one[:24].hex())
        print("  both decode:", z.zstd_decompress(comps[k], **kw) == items[k], z.zstd_decompress(one, **kw) == items[k])
        ndiff = sum(1 for a, b in zip(comps, again) if a != b)
        print("  items differing between two batch calls:", ndiff)
        # consume the same randomness as the fuzzer does for the rest of the round
        for i in rng.integers(0, len(items), 5): pass
    else:
        k = int(rng.integers(0, len(items)))
        for i in rng.integers(0, len(items), 5): pass
    for _ in range(int(rng.integers(1, 7))):
        x = obj(int(rng.choice([3000, 300000, 1500000])))
        r = rng.random()
        if r < 0.35: rng.random(); rng.random()
        elif r < 0.6: rng.random()
        else: rng.choice([1, 3, 6]); rng.random()
        if rng.random() < 0.3:
            rng.integers(0, 256, int(rng.integers(0, 50)), dtype=np.uint8); rng.integers(0, 16)
    # the cut
    # (needs len(buf): not replayed exactly -- rounds after the first wanted one may drift)
    break_after = rd in want_rd
    if break_after and rd == max(want_rd): break
    rng.integers(0, 1 << 20)
