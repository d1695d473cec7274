#!/usr/bin/env python3
"""Host entries on the 1 GiB mix, one frame, with and without the seek table (ZL_c_seekTable): what the table buys the
decoder (sections decoded and downloaded in overlap instead of one frame decoded behind its upload).  usage: host_seek_bench.py"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
n = 1 << 30
host = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy()
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
for seek in (0, 1):
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_setParameter(cctx, 9006, seek)
    best_c = best_d = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n); t1 = time.perf_counter()
        assert not _lib.is_error(r)
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
        assert r2 == n
        best_c, best_d = min(best_c, t1 - t0), min(best_d, t2 - t1)
    assert np.array_equal(back, host)
    print(f"seek table {seek}: {r} bytes; compress {n / best_c / 1e9:.1f} GB/s ({best_c * 1e3:.1f} ms), decompress {n / best_d / 1e9:.1f} GB/s ({best_d * 1e3:.1f} ms)", flush=True)
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
