"""Dev: config C4's compress and decompress calls alone (C-ABI batch entries, 100 000 objects): best of five, the library's own call phases."""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from zstdlite_amd import _lib
from zl_testlib import Ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
items = bench.c4_items(n)
L = _lib.lib()
B = bench.C4Batch(items, L.zl_compressBound)
vp, sp = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
d = Ref().train_dict(items[:2000], 5000)
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
L.zl_CCtx_loadDictionary(cctx, d, len(d)); L.zl_DCtx_loadDictionary(dctx, d, len(d))
PROF = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # (0: the calls as a caller makes them -- packed and unpacked in pieces beside the copies; 1: ZL_c_profile / ZL_d_profile, the phases serialized)
L.zl_CCtx_setParameter(cctx, 9001, PROF); L.zl_DCtx_setParameter(dctx, 9002, PROF)
tc = td = 1e9
for it in range(6):
    t0 = time.perf_counter()
    r = L.zl_compress_batch(cctx, B.n, B.p_comp.ctypes.data_as(vp), B.caps.ctypes.data_as(sp), B.p_src.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.csize.ctypes.data_as(sp))
    t1 = time.perf_counter()
    r2 = L.zl_decompress_batch(dctx, B.n, B.p_back.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.p_comp.ctypes.data_as(vp), B.csize.ctypes.data_as(sp), B.bsize.ctypes.data_as(sp))
    t2 = time.perf_counter()
    assert not _lib.is_error(r) and not _lib.is_error(r2)
    tc, td = min(tc, t1 - t0), min(td, t2 - t1)
ms = (C.c_double * 5)()
cp = dp = None
if PROF:
    L.zl_cctx_batch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; L.zl_cctx_batch_ms(cctx, ms); cp = [round(x, 3) for x in ms]
    L.zl_dctx_batch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; L.zl_dctx_batch_ms(dctx, ms); dp = [round(x, 3) for x in ms]
assert (B.bsize == B.sizes).all()
print(f"compress {tc*1e3:.2f} ms ({n/tc/1e6:.1f} M/s) phases {cp} | decompress {td*1e3:.2f} ms ({n/td/1e6:.1f} M/s) phases {dp} | ratio {B.sizes.sum()/B.csize.sum():.4f}")
