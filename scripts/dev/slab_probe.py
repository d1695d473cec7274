"""ONE frame of 1 GiB (the mix) from host memory to host memory: the slab-by-slab route of zl_decompressDCtx (ZL_d_slabMiB,
include/zl_hip.h) against the whole-frame route (ZL_d_pipeline = 0 keeps the three legs one after the other; ZL_d_slabMiB above
the frame's size keeps the upload early but decodes the frame as a whole), for several run sizes, with and without a content
checksum, and for a frame the reference wrote.  usage: slab_probe.py [runBlocks,...]"""
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
n = 1 << 30
runs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4096, 2048]
host = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy()
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)


def timed(dctx, r, what):
    best = 1e9
    for _ in range(5):
        back[:: 1 << 20] = 0
        t0 = time.perf_counter(); r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t1 = time.perf_counter()
        assert r2 == n, _lib.error_name(r2) if hasattr(_lib, "error_name") else r2
        best = min(best, t1 - t0)
    assert np.array_equal(back, host)
    print(f"{what}: {n / best / 1e9:.1f} GB/s ({best * 1e3:.1f} ms), runs so far {L.zl_dctx_frame_runs(dctx)}", flush=True)


def sweep(r, label):
    others = [] if "only" in sys.argv else [("three legs in a row", {9005: 0}), ("whole frame, early upload", {9010: 4096})]
    for name, params in others + [(f"slabs, runBlocks {b}", {9009: b}) for b in runs]:
        dctx = L.zl_createDCtx()
        for k, v in params.items():
            assert L.zl_DCtx_setParameter(dctx, k, v) == 0
        timed(dctx, r, f"{label}, {name}")
        L.zl_freeDCtx(dctx)


for checksum in (0, 1):
    cctx = L.zl_createCCtx()
    L.zl_CCtx_setParameter(cctx, 201, checksum)
    r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
    assert not _lib.is_error(r)
    L.zl_freeCCtx(cctx)
    sweep(r, f"own frame, checksum {checksum}")
try:
    from zl_testlib import Ref
    ref = Ref()
except Exception as e:  # (oracle/_ref not built)
    print("no reference build:", e)
    sys.exit(0)
for level, ck in ((3, 0), (3, 1)):
    rc = ref.L.ZSTD_createCCtx()
    ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_compressionLevel, level); ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_checksumFlag, ck)
    r = ref.L.ZSTD_compress2(rc, comp.ctypes.data, cap, host.ctypes.data, n)
    ref.L.ZSTD_freeCCtx(rc)
    sweep(r, f"reference frame level {level}, checksum {ck}")
