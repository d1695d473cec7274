"""Host-entry decompression of 1 GiB (independent frames): pageable vs registered host buffers, pipeline on / off."""
import ctypes as C, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
hip = C.CDLL("libamdhip64.so"); hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
n = 1 << 30
host = gen_mixed(n, dev)[0].cpu().numpy()
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8)
back = np.empty(n, dtype=np.uint8); back.fill(1)
for layout in (2, 1):
    cctx = L.zl_createCCtx(); L.zl_CCtx_setParameter(cctx, 9003, layout)
    r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
    for reg in (0, 1):
        if reg: print("register", hip.hipHostRegister(comp.ctypes.data, cap, 0), hip.hipHostRegister(back.ctypes.data, n, 0))
        for pipe in (1, 0):
            dctx = L.zl_createDCtx(); L.zl_DCtx_setParameter(dctx, 9005, pipe)
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter(); r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); best = min(best, time.perf_counter() - t0)
            if r2 != n: print('ERROR', _lib.error_name(r2)); break
            print(f"layout {layout} registered {reg} pipeline {pipe}: {best * 1e3:.1f} ms = {n / best / 1e9:.1f} GB/s, slots {L.zl_dctx_last_slots(dctx)}")
            L.zl_freeDCtx(dctx)
    if reg:
        hip.hipHostUnregister(comp.ctypes.data); hip.hipHostUnregister(back.ctypes.data)
    L.zl_freeCCtx(cctx)
