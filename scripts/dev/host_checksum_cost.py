"""What ZL_c_checksumFlag costs the host entries on the 1 GiB mix (one frame, and one frame with the seek table; a frame the
reference wrote with a checksum): XXH64 is one chain, a CPU core's work, beside the upload (compression) and behind or beside
the download (decompression).  usage: host_checksum_cost.py"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
from zl_testlib import Ref
L = _lib.lib()
n = 1 << 30
host = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy()
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
for seek in (0, 1):
    for checksum in (0, 1):
        cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
        L.zl_CCtx_setParameter(cctx, 9006, seek); L.zl_CCtx_setParameter(cctx, 201, checksum)
        best_c = best_d = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n); t1 = time.perf_counter()
            assert not _lib.is_error(r)
            r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
            assert r2 == n
            best_c, best_d = min(best_c, t1 - t0), min(best_d, t2 - t1)
        assert np.array_equal(back, host)
        print(f"seek table {seek} checksum {checksum}: compress {n / best_c / 1e9:.1f} GB/s ({best_c * 1e3:.1f} ms), decompress {n / best_d / 1e9:.1f} GB/s ({best_d * 1e3:.1f} ms)", flush=True)
        L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
ref = Ref()
for checksum in (0, 1):
    rc = ref.L.ZSTD_createCCtx()
    ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_compressionLevel, 3); ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_checksumFlag, checksum)
    r = ref.L.ZSTD_compress2(rc, comp.ctypes.data, cap, host.ctypes.data, n)
    ref.L.ZSTD_freeCCtx(rc)
    dctx = L.zl_createDCtx()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t1 = time.perf_counter()
        assert r2 == n
        best = min(best, t1 - t0)
    print(f"reference-made frame, checksum {checksum}: decompress {n / best / 1e9:.1f} GB/s ({best * 1e3:.1f} ms)", flush=True)
    L.zl_freeDCtx(dctx)
t0 = time.perf_counter()
from zl_testlib import Oracle
x = Oracle().xxh64(host[: 256 << 20].tobytes())
print(f"(the oracle's XXH64 on one core of this host: {(256 << 20) / (time.perf_counter() - t0) / 1e9:.1f} GB/s incl. a copy)")
