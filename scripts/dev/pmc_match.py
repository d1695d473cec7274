"""Compress one kind of input a few times (for rocprofv3 --pmc passes over the match kernel)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = 256 << 20
codec = DeviceCodec()
src = gen_device(kind, n, dev)
comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
for it in range(3):
    codec.compress(src, comp)
torch.cuda.synchronize()
