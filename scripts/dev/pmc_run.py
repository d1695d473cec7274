#!/usr/bin/env python3
"""Dev: a few compress + decompress passes over one kind of input, no instrumentation: the program under rocprofv3 --pmc.
usage: pmc_run.py [kind=text] [MiB=256] [layout=frames]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
layout = sys.argv[3] if len(sys.argv) > 3 else "frames"
dev = torch.device("cuda", 0)
codec = DeviceCodec(frame_layout=layout)
src = gen_device(kind, n, dev)
comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
for _ in range(3):
    codec.compress(src, comp)
torch.cuda.synchronize()
