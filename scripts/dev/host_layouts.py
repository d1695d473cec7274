"""Dev: host decompress of the 1 GiB mix in the frames layout and one frame + seek table (the slots pipeline)."""
import ctypes as C, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
n = 1 << 30
host = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy()
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
for layout, seek in ((1, 0), (2, 1)):
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_setParameter(cctx, 9003, layout); L.zl_CCtx_setParameter(cctx, 9006, seek)
    r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); best = min(best, time.perf_counter() - t0)
        assert r2 == n
    assert np.array_equal(back, host)
    print(f"layout {layout} seek {seek}: decompress {n / best / 1e9:.1f} GB/s ({best * 1e3:.1f} ms)", flush=True)
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
