"""Host-entry compress / decompress of the 1 GiB mix, several contexts in one process: how often do the copy streams overlap?"""
import ctypes as C, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed, DeviceCodec
L = _lib.lib()
L.zl_cctx_copy_duplex.argtypes = [C.c_void_p]; L.zl_cctx_copy_duplex.restype = C.c_int
dev = torch.device("cuda", 0)
n = 1 << 30
src = gen_mixed(n, dev)[0]
host = src.cpu().numpy()
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
for trial in range(4):
    if trial == 2:  # what bench.py has behind it by then: device-resident round trips, torch copies
        c = DeviceCodec(); cb = torch.empty(c.bound(n), dtype=torch.uint8, device=dev)
        for _ in range(2): c.compress(src, cb)
        _ = src.cpu()
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n); t1 = time.perf_counter()
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
        assert r2 == n
        ts.append((t1 - t0, t2 - t1))
    print(f"context {trial}: duplex {L.zl_cctx_copy_duplex(cctx)} compress ms {[round(a * 1e3, 1) for a, _ in ts]} decompress ms {[round(b * 1e3, 1) for _, b in ts]}", flush=True)
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
assert np.array_equal(back, host)
