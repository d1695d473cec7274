"""Dev: the one-frame layout's ratio on inputs the table parse takes (above 16 MiB): records 64 MiB, text 64 MiB, corpus x 3."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
for name, data in (("records64M", LONG_RANGE["records"](64 << 20)), ("text64M", GENERATORS["text"](64 << 20)), ("corpus_x3", LONG_RANGE["corpus"](None) * 3)):
    comp = z.zstd_compress(data)
    theirs = len(ref.compress(data, level=3))
    print(f"{name}: ours {len(data)/len(comp):.3f} ref {len(data)/theirs:.3f} size ratio {len(comp)/theirs:.4f}", flush=True)
