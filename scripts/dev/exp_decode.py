"""Timing-only decode experiment: compress with the product library, decode with whatever libzl_hip.so is in place, ignore errors."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, ctypes as C
from zstdlite_amd.device import DeviceCodec, gen_device
from zstdlite_amd import _lib
dev = torch.device("cuda", 0)
n = 256 << 20
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
codec = DeviceCodec(profile=2)
src = gen_device(kind, n, dev)
comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
back = torch.empty(n, dtype=torch.uint8, device=dev)
_, total, fs = codec.compress(src, comp)
tbl = codec.frame_table(fs, codec.chunk_content_sizes(n))
best = {}
for it in range(4):
    try:
        codec.decompress(comp, tbl, back)
    except Exception as e:
        pass
    for k, v in codec.kernel_ms().items():
        if k.startswith("d"):
            best[k] = min(best.get(k, 1e9), v)
st = (C.c_ulonglong * 16)()
codec.L.zl_dctx_phase_stats(C.c_void_p(codec.dctx), st)
nwg = n / (128 << 10) / 8
print(kind, "ok" if torch.equal(back, src) else "MISMATCH", {k: round(v, 3) for k, v in best.items()}, f"roles kcycles/wg: lit {st[14]/nwg/1e3:.0f} seq {st[15]/nwg/1e3:.0f}  seq build {st[0]/max(st[2],1)/1e3:.0f} chain {st[1]/max(st[2],1)/1e3:.0f} parse {st[3]/max(st[2],1)/1e3:.0f} tables {st[4]/max(st[2],1)/1e3:.0f}  exec kcycles/frame {st[5]/(n/(128<<10))/1e3:.0f} (prologue {st[6]/(n/(128<<10))/1e3:.0f}, final drain {st[7]/(n/(128<<10))/1e3:.0f})")
