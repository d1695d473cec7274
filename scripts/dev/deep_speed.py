#!/usr/bin/env python3
"""Dev: the deep levels on inputs that fill the device, device-resident: per-stage milliseconds (search = key + sort + search +
extension; parse = zl_match_deep_kernel; entropy) and the compressed size."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
dev = torch.device("cuda:0")
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 6, 9, 12]
inputs = {"text256M": gen_device("text", 256 << 20, dev), "mix1G": gen_mixed(1 << 30, dev)[0]}
for name, src in inputs.items():
    for layout in ("one_frame", "frames"):
        for lvl in levels:
            codec = DeviceCodec(level=lvl, profile=True, frame_layout=layout)
            dst = torch.empty(codec.bound(src.numel()), dtype=torch.uint8, device=dev)
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.time()
                _, total, fs = codec.compress(src, dst)
                torch.cuda.synchronize(); dt = time.time() - t0
            ms = codec.kernel_ms()
            print(f"{name} {layout:9s} level {lvl:2d}: ratio {src.numel()/total:6.3f}  wall {dt*1e3:7.1f} ms = {src.numel()/dt/1e9:6.2f} GB/s   search/far {ms.get('far', 0):7.2f}  parse {ms.get('match', 0):6.2f}  entropy {ms.get('entropy', 0):5.2f} ms", flush=True)
            codec.close()
