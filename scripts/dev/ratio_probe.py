import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import GENERATORS, Ref
ref = Ref()
levels = [int(x) for x in sys.argv[1:]] or [3]
for level in levels:
    for gen in ("text", "iid10", "int20"):
        data = GENERATORS[gen](8 << 20)
        ours = len(z.zstd_compress(data, level=level)); theirs = len(ref.compress(data, level=level))
        print("level", level, gen, ours, theirs, round(ours / theirs, 4))
