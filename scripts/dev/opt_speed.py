"""Dev: wall time of the 1 GiB bench vector at the deep / price-based levels, device-resident, one frame."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
dev = torch.device("cuda", 0)
n = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else 1 << 30
src, _, _ = gen_mixed(n, dev)
for level in [int(x) for x in sys.argv[1].split(",")]:
    c = DeviceCodec(level=level, frame_layout="one_frame", profile=True)
    comp = torch.empty(c.bound(n), dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        _, total, fs = c.compress(src, comp)
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    back = torch.empty_like(src)
    print(f"level {level}: {best * 1e3:.1f} ms  {n / best / 1e9:.2f} GB/s  ratio {n / total:.4f}  kernels {c.kernel_ms()}", flush=True)
    c.close()
