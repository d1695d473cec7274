#!/usr/bin/env python3
"""Dev: the encoder's phase counters (ZL_c_profile 2) over a C4-like batch of small objects with a dictionary."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import bench  # noqa: E402
from zstdlite_amd import _lib  # noqa: E402
from zl_testlib import Ref  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
use_dict = (sys.argv[2] if len(sys.argv) > 2 else "dict") == "dict"
items = bench.c4_items(n)
L = _lib.lib()
B = bench.C4Batch(items, L.zl_compressBound)
vp, sp = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
cctx = L.zl_createCCtx()
if use_dict:
    d = Ref().train_dict(items[:2000], 5000)
    L.zl_CCtx_loadDictionary(cctx, d, len(d))
L.zl_CCtx_setParameter(cctx, 9001, 2)
L.zl_cctx_batch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
for it in range(3):
    r = L.zl_compress_batch(cctx, B.n, B.p_comp.ctypes.data_as(vp), B.caps.ctypes.data_as(sp), B.p_src.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.csize.ctypes.data_as(sp))
    assert not _lib.is_error(r)
ms = (C.c_double * 5)()
L.zl_cctx_batch_ms(cctx, ms)
print("call phases ms", [round(x, 3) for x in ms])
km = (C.c_float * 8)()
if hasattr(L, "zl_cctx_kernel_ms"):
    L.zl_cctx_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
ct = (C.c_ulonglong * 16)()
L.zl_cctx_phase_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert L.zl_cctx_phase_stats(cctx, ct) == 0
if ct[4]:
    it = ct[4]
    print(f"match phases, cycles/iteration: find {ct[0]/it:.0f} parse {ct[1]/it:.0f} emit {ct[2]/it:.0f} insert {ct[3]/it:.0f}  iterations/object {it/n:.1f}  sequences/object {ct[5]/n:.1f}")
if ct[14]:
    nc = ct[14]
    print(f"entropy phases, kcycles/object: statistics {ct[8]/nc/1e3:.1f}  tree|tables {ct[9]/nc/1e3:.1f}  codes+bits {ct[10]/nc/1e3:.1f}  chains|tree description (+tail) {ct[13]/nc/1e3:.1f}  blk literals {ct[11]/nc/1e3:.1f}  blk sequences {ct[12]/nc/1e3:.1f}")
print("ratio", B.sizes.sum() / B.csize.sum())

# the decoder's side: the general kernel's phase counters (ZL_d_profile 2)
dctx = L.zl_createDCtx()
if use_dict:
    L.zl_DCtx_loadDictionary(dctx, d, len(d))
L.zl_DCtx_setParameter(dctx, 9002, 2)
L.zl_dctx_batch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
for it in range(3):
    r = L.zl_decompress_batch(dctx, B.n, B.p_back.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.p_comp.ctypes.data_as(vp), B.csize.ctypes.data_as(sp), B.bsize.ctypes.data_as(sp))
    assert not _lib.is_error(r)
L.zl_dctx_batch_ms(dctx, ms)
print("decode call phases ms", [round(x, 3) for x in ms])
st = (C.c_ulonglong * 16)()
L.zl_dctx_phase_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
if L.zl_dctx_phase_stats(dctx, st) == 0 and st[5]:
    nb = st[5]
    print(f"decode phases, kcycles/block: H {st[0]/nb/1e3:.1f}  B {st[1]/nb/1e3:.1f}  A {st[2]/nb/1e3:.1f}  A||B {st[3]/nb/1e3:.1f}  C {st[4]/nb/1e3:.1f}   blocks {nb}  seq/block {st[6]/nb:.0f}  huflits/block {st[7]/nb:.0f}")
else:
    print("no decode phase stats", list(st))
