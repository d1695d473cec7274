"""Dev: the host entries on the independent-frames layout (1 GiB mix), with the dev library's ZL_DEBUG_SLAB lines."""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
layout = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = _lib.lib()
host = gen_mixed(1 << 30, torch.device("cuda", 0))[0].cpu().numpy()
n = host.size
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
L.zl_CCtx_setParameter(cctx, 9003, layout)
for it in range(4):
    t0 = time.perf_counter(); r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n); t1 = time.perf_counter()
    print(f"--- iteration {it}: compress {(t1 - t0) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
    print(f"--- iteration {it}: decompress {(t2 - t1) * 1e3:.1f} ms = {n / (t2 - t1) / 1e9:.1f} GB/s", file=sys.stderr, flush=True)
assert r2 == n and np.array_equal(back, host)
