"""Dev: the "table" input of tests/test_compress_gpu.py (400 lines of numbers as text, drawn at random) by level, against the reference."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import numpy as np
import zstdlite_amd as z
from zl_testlib import Ref
ref = Ref()
rng = np.random.default_rng(11)
iid = None
_ = rng.integers(48, 58, size=3 << 20, dtype=np.uint8)
lines = [b",".join(b"%d" % v for v in rng.integers(0, 1000, size=12)) + b"\n" for _ in range(400)]
table = b"".join(lines[i] for i in rng.integers(0, 400, size=60000))
for lv in (1, 3, 4, 5, 6, 9):
    theirs = len(ref.compress(table, level=lv))
    for lay in ("auto", "frames"):
        ours = len(z.zstd_compress(table, level=lv, frame_layout=lay))
        print(len(table), "level", lv, lay, "ours/ref %.3f" % (ours / theirs))
