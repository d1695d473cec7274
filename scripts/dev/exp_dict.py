"""C4 shape: many small objects compressed with a trained dictionary.  Checks the round trip through the
reference decoder and ours, and prints ratio and throughput next to the reference library's."""
import sys, time, pickle
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zl_testlib import Ref, Oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(3)
names = [f"name_{i:02d}_{'x' * (i % 7)}" for i in range(50)]
def obj(i):
    perm = rng.permutation(50)[: int(rng.integers(10, 40))]
    return pickle.dumps({names[j]: int(rng.integers(0, 1000)) for j in perm}, protocol=4)
items = [obj(i) for i in range(N)]
ref = Ref(); orc = Oracle()
d = ref.train_dict(items[:2000], 5000)
print("dict", len(d), "bytes, id", z.zstd_dict_id(d), "items", N, "avg size", sum(map(len, items)) / N)
raw = sum(map(len, items))
t0 = time.time(); plain = z.zstd_compress_batch(items); t_plain = time.time() - t0
cctx = z.zstd_cctx(dict=d)
t0 = time.time(); withd = z.zstd_compress_batch(items, cctx=cctx); t_dict = time.time() - t0
t0 = time.time(); withd = z.zstd_compress_batch(items, cctx=cctx); t_dict = time.time() - t0
t0 = time.time(); cpu = [ref.compress(x, level=3, dict=d) for x in items[:2000]]; t_cpu = (time.time() - t0) * N / 2000
print(f"ours no dict: ratio {raw / sum(map(len, plain)):.3f} ({t_plain:.2f}s)   ours dict: ratio {raw / sum(map(len, withd)):.3f} ({t_dict:.2f}s, {N / t_dict:.0f} items/s)   cpu dict: ratio {sum(map(len, items[:2000])) / sum(map(len, cpu)):.3f} (~{t_cpu:.2f}s for all)")
# parity: reference decoder and oracle with the dictionary, ours with the dictionary
for i in range(0, N, max(1, N // 300)):
    assert ref.decompress(withd[i], len(items[i]), dict=d) == items[i], i
    assert orc.decompress(withd[i], cap=len(items[i]), dict=d) == items[i], i
back = z.zstd_decompress_batch(withd, dict=d)
assert back == items
print("round trips ok; dict id in frame:", z.zstd_dict_id(withd[0]))
