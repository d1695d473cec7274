#!/usr/bin/env python3
"""Dev: per-stage milliseconds of a small device-resident input at levels 3 and 6 (ZL_c_profile events)."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device
dev = torch.device("cuda:0")
levels = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 6]
for kind in ("text", "int20"):
    for n in (64 << 10, 1 << 20, 4 << 20, 16 << 20):
        src = gen_device(kind, n, dev)
        for lvl in levels:
            codec = DeviceCodec(level=lvl, profile=True, frame_layout="one_frame")
            dst = torch.empty(codec.bound(src.numel()), dtype=torch.uint8, device=dev)
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.time()
                _, total, fs = codec.compress(src, dst)
                torch.cuda.synchronize(); best = min(best, time.time() - t0)
            ms = codec.kernel_ms()
            print(f"{kind:6s} {n:9d} level {lvl:2d}: ratio {src.numel()/total:6.3f}  wall {best*1e3:7.3f} ms   search/far {ms.get('far', 0):7.3f}  parse {ms.get('match', 0):6.3f}  entropy {ms.get('entropy', 0):5.3f} ms", flush=True)
            codec.close()
