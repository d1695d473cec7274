#!/usr/bin/env python3
"""Dev: the general decode kernel on frames of 64 KiB that carry a dictionary (so that the lane pipeline does not take them):
kernel phase of zl_decompress_batch.  usage: general_bench.py [items=4000]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import bench
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
text = GENERATORS["text"](n * 65536 + 65536, seed=11)
items = [text[i * 65536:(i + 1) * 65536] for i in range(n)]
d = Ref().train_dict(items[:200], 20000)
L = _lib.lib()
B = bench.C4Batch(items, L.zl_compressBound)
vp, sp = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
L.zl_CCtx_loadDictionary(cctx, d, len(d)); L.zl_DCtx_loadDictionary(dctx, d, len(d))
L.zl_DCtx_setParameter(dctx, 9002, 1)
L.zl_dctx_batch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
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
print(f"{n} frames of 64 KiB with a dictionary: general kernel phase {best:.3f} ms = {n * 65536 / best / 1e6:.1f} GB/s; ratio {B.sizes.sum() / B.csize.sum():.3f}")
