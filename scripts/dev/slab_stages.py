"""Stage times of the runs of ONE slab-by-slab decode (a ZL_DEV_SWITCHES build with ZL_DEBUG_BIG=1 prints them to stderr).
usage: ZL_HIP_LIB=...dev.so ZL_DEBUG_BIG=1 slab_stages.py [runBlocks]"""
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
cctx = L.zl_createCCtx()
r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
import os
if os.environ.get("REF"):  # a frame the reference wrote (level REF) instead of this library's
    from zl_testlib import Ref
    ref = Ref()
    rc = ref.L.ZSTD_createCCtx()
    ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_compressionLevel, int(os.environ["REF"]))
    r = ref.L.ZSTD_compress2(rc, comp.ctypes.data, cap, host.ctypes.data, n)
    ref.L.ZSTD_freeCCtx(rc)
if os.environ.get("FREE_CCTX"):
    L.zl_freeCCtx(cctx)
dctx = L.zl_createDCtx()
if len(sys.argv) > 1:
    L.zl_DCtx_setParameter(dctx, 9009, int(sys.argv[1]))
for k in range(6):
    print(f"--- call {k}", file=sys.stderr, flush=True)
    if os.environ.get("TOUCH"):
        back[:: 1 << 20] = 0
    t0 = time.perf_counter(); r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t1 = time.perf_counter()
    print(f"--- call {k}: {(t1 - t0) * 1e3:.1f} ms", file=sys.stderr, flush=True)
assert r2 == n and np.array_equal(back, host)
