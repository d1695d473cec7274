"""Dev: level-6 ratio and match time in the independent-frames layout (the HI instantiation without INTRA)."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import torch, numpy as np
import zstdlite_amd as z
from zstdlite_amd.device import DeviceCodec
from zl_testlib import GENERATORS, LONG_RANGE
ins = {"corpus": LONG_RANGE["corpus"](None), "text": GENERATORS["text"](16 << 20)}
dev = torch.device("cuda", 0)
for level in (5, 6):
    for k, d in ins.items():
        c = z.zstd_compress(d, level=level, frame_layout="frames")
        assert z.zstd_decompress(c) == d
        codec = DeviceCodec(level=level, profile=True)
        src = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).to(dev)
        for _ in range(3): codec.compress(src)
        print(f"level {level} {k:8s} ratio {len(d)/len(c):7.4f}  match {codec.kernel_ms()['match']:.3f} ms", flush=True)
