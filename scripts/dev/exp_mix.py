import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
src0, _, _ = gen_mixed(n, dev)
for name, src in (("contiguous", src0), ("interleaved", src0.view(4, -1, 128 << 10).transpose(0, 1).contiguous().view(-1))):
    codec = DeviceCodec(profile=True)
    comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    best = {}
    for it in range(4):
        _, total, fs = codec.compress(src, comp)
        tbl = codec.frame_table(fs, codec.chunk_content_sizes(n))
        codec.decompress(comp, tbl, back)
        for k, v in codec.kernel_ms().items():
            best[k] = min(best.get(k, 1e9), v)
    assert torch.equal(back, src)
    print(name, n >> 20, "MiB", {k: round(v, 2) for k, v in best.items() if k in ("match", "entropy", "decode", "d_entropy", "d_exec")})
