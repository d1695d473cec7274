"""Device-resident timings of the two frame layouts on one input: kernels of the encoder (far passes, match,
entropy, gather) and the decoder, ratio.  usage: python scripts/layout_bench.py [mix|corpus|records|text|...] [MiB]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "mix"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = mib << 20
if kind == "mix":
    src = gen_mixed(n, dev)[0]
elif kind in ("corpus", "records"):
    from zl_testlib import LONG_RANGE
    src = torch.from_numpy(np.frombuffer(LONG_RANGE[kind](n), dtype=np.uint8).copy()).to(dev)
else:
    src = gen_device(kind, n, dev)
for layout in ("frames", "one_frame"):
    codec = DeviceCodec(profile=True, frame_layout=layout)
    comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    best, tbest_c, tbest_d = {}, 1e9, 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, total, fs = codec.compress(src, comp)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tbl = codec.frame_table(fs, codec.chunk_content_sizes(n)) if layout == "frames" else codec.frame_table([total], [n])
        back.zero_()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        codec.decompress(comp, tbl, back, chunk_sizes=fs if layout == "one_frame" else None)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if it:
            tbest_c = min(tbest_c, t1 - t0); tbest_d = min(tbest_d, t3 - t2)
        for k, v in codec.kernel_ms().items():
            best[k] = min(best.get(k, 1e9), v)
    assert torch.equal(back, src)
    print(f"{kind} {mib} MiB {layout:9s} ratio {n/total:7.3f} | compress {tbest_c*1e3:7.2f} ms ({n/tbest_c/1e9:6.1f} GB/s) decompress {tbest_d*1e3:7.2f} ms ({n/tbest_d/1e9:6.1f} GB/s) | "
          + " ".join(f"{k} {v:.3f}" for k, v in best.items()), flush=True)
