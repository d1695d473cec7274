#!/usr/bin/env python3
"""Dev: one configuration of the deep levels, three compressions (for rocprofv3 --kernel-trace --stats)."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
dev = torch.device("cuda:0")
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 9
kind = sys.argv[2] if len(sys.argv) > 2 else "mix"
src = gen_mixed(1 << 30, dev)[0] if kind == "mix" else gen_device(kind, 256 << 20, dev)
codec = DeviceCodec(level=lvl, frame_layout="one_frame")
dst = torch.empty(codec.bound(src.numel()), dtype=torch.uint8, device=dev)
for _ in range(3):
    _, total, fs = codec.compress(src, dst)
torch.cuda.synchronize()
print("ratio", src.numel() / total)
