"""Dev: zl_decompressDCtx of this library's own 1 GiB one-frame output from host memory through ctypes, with ZL_DEBUG_SLAB's timeline (stderr)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
host = gen_mixed(1 << 30, torch.device("cuda", 0))[0].cpu().numpy()
n = host.size
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
for it in range(4):
    t0 = time.perf_counter()
    r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
    dt = time.perf_counter() - t0
    print(f"--- iteration {it}: {dt*1e3:.1f} ms  {n/dt/1e9:.1f} GB/s", file=sys.stderr, flush=True)
assert r2 == n and np.array_equal(back, host)
