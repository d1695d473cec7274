"""Dev: config C4's batch calls alone (100 000 small objects, trained dictionary, C-ABI), for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 scripts/dev/c4_prof.py"""
import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from bench import c4_items, C4Batch
from zstdlite_amd import _lib
from zl_testlib import Ref
ref = Ref(); L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
items = c4_items(n)
d = ref.train_dict(items[:2000], 5000)
B = C4Batch(items, L.zl_compressBound)
vp = C.POINTER(C.c_void_p); sp = C.POINTER(C.c_size_t)
for fn in ("zl_cctx_batch_ms", "zl_dctx_batch_ms"):
    getattr(L, fn).argtypes = [C.c_void_p, C.POINTER(C.c_double)]; getattr(L, fn).restype = C.c_size_t
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
L.zl_CCtx_loadDictionary(cctx, d, len(d)); L.zl_DCtx_loadDictionary(dctx, d, len(d))
L.zl_CCtx_setParameter(cctx, 9001, 1); L.zl_DCtx_setParameter(dctx, 9002, 1)
cms, dms = (C.c_double * 5)(), (C.c_double * 5)()
for it in range(4):
    t1 = time.perf_counter()
    r = L.zl_compress_batch(cctx, B.n, B.p_comp.ctypes.data_as(vp), B.caps.ctypes.data_as(sp), B.p_src.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.csize.ctypes.data_as(sp))
    t2 = time.perf_counter()
    r2 = L.zl_decompress_batch(dctx, B.n, B.p_back.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.p_comp.ctypes.data_as(vp), B.csize.ctypes.data_as(sp), B.bsize.ctypes.data_as(sp))
    t3 = time.perf_counter()
    L.zl_cctx_batch_ms(cctx, cms); L.zl_dctx_batch_ms(dctx, dms)
    print(f"compress {(t2 - t1) * 1e3:.2f} ms {[round(x, 2) for x in cms]}   decompress {(t3 - t2) * 1e3:.2f} ms {[round(x, 2) for x in dms]}", flush=True)
assert (B.bsize == B.sizes).all() and np.array_equal(B.back, B.blob)
