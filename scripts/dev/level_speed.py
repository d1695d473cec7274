"""Kernel times of the 1 GiB bench vector per level and layout (HIP events inside the library)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
dev = torch.device("cuda", 0)
n = 1 << 30
src, _, _ = gen_mixed(n, dev)
for level in (1, 3, 5):
    for layout in ("frames", "one_frame"):
        c = DeviceCodec(level=level, frame_layout=layout, profile=True)
        comp = torch.empty(c.bound(n), dtype=torch.uint8, device=dev)
        best = {}
        for _ in range(3):
            _, total, fs = c.compress(src, comp)
            for k, v in c.kernel_ms().items():
                if k in ("match", "entropy", "far", "gather"):
                    best[k] = min(best.get(k, 1e9), v)
        tot = sum(best.values())
        print(f"level {level} {layout:9s} ratio {n / total:.4f}  " + "  ".join(f"{k} {v:.2f}" for k, v in best.items()) + f"  -> {n / tot / 1e6:.0f} GB/s (kernels)")
        c.close()
