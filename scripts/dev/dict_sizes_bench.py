#!/usr/bin/env python3
"""Dev: decompress_batch of dictionary frames of one content size, kernel phase per frame: where the lane pipeline overtakes the
general kernel (ZL_DICT_LANE_MIN).  usage: dict_sizes_bench.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import bench
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref
L = _lib.lib()
ref = Ref()
text = GENERATORS["text"](64 << 20, seed=3)
d = ref.train_dict([text[i * 5000:(i + 1) * 5000] for i in range(400)], 20000)
vp, sp = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
L.zl_dctx_batch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
for size in (512, 1024, 2048, 4096, 8192, 16384):
    n = min(40000, (48 << 20) // size)
    items = [text[(2 << 20) + i * size:(2 << 20) + (i + 1) * size] for i in range(n)]
    B = bench.C4Batch(items, L.zl_compressBound)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_loadDictionary(cctx, d, len(d)); L.zl_DCtx_loadDictionary(dctx, d, len(d))
    L.zl_DCtx_setParameter(dctx, 9002, 1)
    r = L.zl_compress_batch(cctx, B.n, B.p_comp.ctypes.data_as(vp), B.caps.ctypes.data_as(sp), B.p_src.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.csize.ctypes.data_as(sp))
    assert not _lib.is_error(r)
    best = 1e9
    ms = (C.c_double * 5)()
    for it in range(4):
        r = L.zl_decompress_batch(dctx, B.n, B.p_back.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.p_comp.ctypes.data_as(vp), B.csize.ctypes.data_as(sp), B.bsize.ctypes.data_as(sp))
        assert not _lib.is_error(r)
        L.zl_dctx_batch_ms(dctx, ms)
        best = min(best, ms[2])
    assert np.array_equal(B.back, B.blob)
    print(f"{n} frames of {size} bytes: kernels {best:.3f} ms = {best * 1e6 / n:.0f} ns per frame, {n * size / best / 1e6:.1f} GB/s", flush=True)
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
