"""Dev: the one-frame layout at level 3 on inputs that do and do not repeat further back: compressed bytes and stage times
(run once per library: ZL_HIP_LIB=... for the variant built with -DZL_FAR_GATE=0)."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
from zl_testlib import LONG_RANGE, GENERATORS
dev = torch.device("cuda:0")
def up(b): return torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev)
inputs = {"mix1G": gen_mixed(1 << 30, dev)[0], "text256M": gen_device("text", 256 << 20, dev), "rand256M": gen_device("rand", 256 << 20, dev),
          "corpus": up(LONG_RANGE["corpus"](None) * 3), "records64M": up(LONG_RANGE["records"](64 << 20)),
          "blob_twice": torch.cat([gen_device("rand", 1 << 20, dev)] * 2 + [gen_device("text", 3 << 20, dev)] + [gen_device("rand", 1 << 20, dev)])}
for lvl in (3, 5):
    for name, src in inputs.items():
        codec = DeviceCodec(level=lvl, profile=True, frame_layout="one_frame")
        dst = torch.empty(codec.bound(src.numel()), dtype=torch.uint8, device=dev)
        for _ in range(3):
            _, total, fs = codec.compress(src, dst)
        ms = codec.kernel_ms()
        print(f"{name:11s} level {lvl}: {total:11d} bytes   far {ms.get('far', 0):6.3f}  match {ms.get('match', 0):6.3f}  entropy {ms.get('entropy', 0):6.3f} ms", flush=True)
        codec.close()
