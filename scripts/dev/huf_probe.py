"""Which literal distributions end up as raw literals?  (development)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "scripts")
import numpy as np
import zstdlite_amd as z
from frame_stats import show
from zl_testlib import Ref
from ratio_inputs import INPUTS
ref = Ref()
rng = np.random.default_rng(1)
def skew(n, k, p, rest=256):
    a = rng.integers(0, rest, n).astype(np.uint8)
    m = rng.random(n) < p
    a[m] = (rng.integers(0, k, int(m.sum())) * (256 // k)).astype(np.uint8)
    return a.tobytes()
cases = {"4sym_p80_rest256": skew(131072, 4, 0.8), "4sym_p80_rest64": skew(131072, 4, 0.8, 64), "16sym_p90": skew(131072, 16, 0.9), "2sym_p95": skew(131072, 2, 0.95),
         "4sym_p99": skew(131072, 4, 0.99), "16sym_only": skew(131072, 16, 1.0)}
d = INPUTS["f64_seq"](1 << 20)
for c in range(8):
    cases[f"f64_seq_chunk{c}"] = d[c * 131072:(c + 1) * 131072]
for name, data in cases.items():
    print("==", name)
    show("ours", z.zstd_compress(data), len(data))
    show("ref", ref.compress(data, 3), len(data))
