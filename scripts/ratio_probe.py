import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import GENERATORS, Ref
ref = Ref()
for gen in ("text", "iid10", "int20"):
    data = GENERATORS[gen](8 << 20)
    ours = len(z.zstd_compress(data)); theirs = len(ref.compress(data, level=3))
    print(gen, ours, theirs, round(ours / theirs, 4))
