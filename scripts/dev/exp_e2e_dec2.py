"""Does host-entry decompression keep its two copy engines over a sequence of calls of different kinds?"""
import ctypes as C, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
dev = torch.device("cuda", 0)
n = 1 << 30
host = gen_mixed(n, dev)[0].cpu().numpy()
cap = L.zl_compressBound(n)
comps = {}
for layout in (1, 2):
    cctx = L.zl_createCCtx(); L.zl_CCtx_setParameter(cctx, 9003, layout)
    buf = np.empty(cap, dtype=np.uint8)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n); ts.append((time.perf_counter() - t0) * 1e3)
    comps[layout] = (buf, r)
    print(f"compress layout {layout}:", " ".join(f"{t:.1f}" for t in ts), "ms")
    L.zl_freeCCtx(cctx)
back = np.empty(n, dtype=np.uint8); back.fill(1)
for step, layout in enumerate((1, 1, 2, 1, 2, 1)):
    buf, r = comps[layout]
    dctx = L.zl_createDCtx()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, buf.ctypes.data, r); ts.append((time.perf_counter() - t0) * 1e3)
    assert r2 == n, _lib.error_name(r2)
    print(f"decompress step {step} layout {layout}:", " ".join(f"{t:.1f}" for t in ts), "ms")
    L.zl_freeDCtx(dctx)
