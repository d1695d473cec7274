"""Dev: ratios in the independent-frames layout (the bulk match instantiation) on real and synthetic inputs."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE
from ratio_inputs import INPUTS
ins = {"corpus": LONG_RANGE["corpus"](None), "records": LONG_RANGE["records"](16 << 20), "text": GENERATORS["text"](8 << 20)}
ins.update({k: g(1 << 20) for k, g in INPUTS.items()})
for k, d in ins.items():
    c = z.zstd_compress(d, frame_layout="frames")
    assert z.zstd_decompress(c) == d
    print(f"{k:16s} {len(d)/len(c):8.4f}", flush=True)
