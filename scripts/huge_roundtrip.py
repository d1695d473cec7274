"""Inputs above 4 GiB (32-bit positions and sizes are the classic casualty): the host entries in both layouts, with a seek table
and device slots, the device entries, and a frame the reference wrote, on N GiB of the bench mix: reference decode (one frame: as a
whole; frames: all of them), our decode, bit-exact.  usage: huge_roundtrip.py [GiB=4.5]"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
from zstdlite_amd import _lib
from zstdlite_amd.device import DeviceCodec, gen_mixed
from zl_testlib import Ref
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.5
n = int(gib * (1 << 30)) + 12345
L = _lib.lib()
ref = Ref()
dev = torch.device("cuda", 0)
t0 = time.time()
src = gen_mixed(n, dev)[0]
host = src.cpu().numpy()
print(f"input {n} bytes ready in {time.time() - t0:.1f}s", flush=True)
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8)
back = np.empty(n, dtype=np.uint8)
for layout, seek, slots in ((0, 0, 1), (1, 0, 1), (2, 1, 3), (1, 1, 2)):
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_setParameter(cctx, 9003, layout); L.zl_CCtx_setParameter(cctx, 9006, seek); L.zl_CCtx_setParameter(cctx, 9007, slots)
    L.zl_DCtx_setParameter(dctx, 9008, slots)
    for rep in range(2):  # (the first call of a context allocates its device and pinned buffers -- gigabytes here; the second is the rate)
        t1 = time.time()
        r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
        assert not _lib.is_error(r), _lib.error_name(r)
        t2 = time.time()
        back.fill(0)
        t2b = time.time()
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
        t3 = time.time()
        if rep == 0:
            first = (n / (t2 - t1) / 1e9, n / (t3 - t2b) / 1e9)
    assert r2 == n, _lib.error_name(r2) if _lib.is_error(r2) else r2
    assert np.array_equal(back, host), "our decoder: bytes differ"
    ok_ref = ref.decompress(comp[:r].tobytes(), n) == host.tobytes()
    assert ok_ref, "reference decoder: bytes differ"
    print(f"layout {layout} seek {seek} slots {slots}: {r} bytes, compress {n / (t2 - t1) / 1e9:.1f} GB/s, decompress {n / (t3 - t2b) / 1e9:.1f} GB/s (first call of the contexts: {first[0]:.1f} / {first[1]:.1f}), reference decode ok", flush=True)
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
# device-resident entries
for layout in ("frames", "one_frame"):
    codec = DeviceCodec(frame_layout=layout)
    dst, total, fs = codec.compress(src)
    assert int(np.sum(fs, dtype=np.uint64)) == total
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    if layout == "frames":
        codec.decompress(dst, codec.frame_table(fs, codec.chunk_content_sizes(n)), out)
    else:
        codec.decompress(dst, codec.frame_table([total], [n]), out, chunk_sizes=fs)
    assert torch.equal(out, src), "device decode differs"
    out.zero_()
    torch.cuda.synchronize()
    t1 = time.time()
    if layout == "frames":
        codec.decompress(dst, codec.frame_table(fs, codec.chunk_content_sizes(n)), out)
    else:
        codec.decompress(dst, codec.frame_table([total], [n]), out, chunk_sizes=fs)
    t2 = time.time()
    assert torch.equal(out, src), "device decode differs"
    print(f"device entries, {layout}: {total} bytes, round trip bit-exact, decompress {n / (t2 - t1) / 1e9:.1f} GB/s", flush=True)
    codec.close()
    del dst, out
# a frame the REFERENCE wrote (ZSTD_compress2, level 3, checksum): one frame, no hint, no seek table -- decoded in runs of blocks
rc = ref.L.ZSTD_createCCtx()
ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_compressionLevel, 3); ref.L.ZSTD_CCtx_setParameter(rc, ref.ZSTD_c_checksumFlag, 1)
t1 = time.time()
r = ref.L.ZSTD_compress2(rc, comp.ctypes.data, cap, host.ctypes.data, n)
t2 = time.time()
ref.L.ZSTD_freeCCtx(rc)
assert not ref.L.ZSTD_isError(r)
dctx = L.zl_createDCtx()
for rep in range(2):
    back.fill(0)
    t3 = time.time()
    r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
    t4 = time.time()
    assert r2 == n, _lib.error_name(r2) if _lib.is_error(r2) else r2
    assert np.array_equal(back, host), "reference frame: bytes differ"
    print(f"reference-made frame: {r} bytes (reference compress {n / (t2 - t1) / 1e9:.2f} GB/s), our decompress {n / (t4 - t3) / 1e9:.1f} GB/s in {int(L.zl_dctx_frame_runs(dctx))} runs so far", flush=True)
rd = ref.L.ZSTD_createDCtx()
t5 = time.time()
r3 = ref.L.ZSTD_decompressDCtx(rd, back.ctypes.data, n, comp.ctypes.data, r)
t6 = time.time()
ref.L.ZSTD_freeDCtx(rd)
assert r3 == n
print(f"reference decompress of the same frame: {n / (t6 - t5) / 1e9:.2f} GB/s", flush=True)
L.zl_freeDCtx(dctx)
print("ok")
