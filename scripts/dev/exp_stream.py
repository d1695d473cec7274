"""Streaming entries (N3) on host buffers: zl_compressStream2 / zl_decompressStream against the reference
library's streaming calls on one host core.  usage: exp_stream.py [MiB]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import zstdlite_amd as z
from zstdlite_amd import stream
from zstdlite_amd.device import gen_mixed
from zl_testlib import Ref
from test_stream_gpu import ref_stream_compress, ref_stream_decompress
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
data = gen_mixed(mb << 20, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
cctx, dctx = z.zstd_cctx(), z.zstd_dctx()
for it in range(2):
    t0 = time.time(); c = stream.Compressor(cctx._ptr, pledged=len(data))
    parts = [c.feed(data[o:o + stream.IN_CHUNK]) for o in range(0, len(data), stream.IN_CHUNK)] + [c.finish()]
    t_c = time.time() - t0
blob = b"".join(parts)
for it in range(2):
    t0 = time.time(); d = stream.Decompressor(dctx._ptr)
    out = [d.feed(blob[o:o + stream.IN_CHUNK]) for o in range(0, len(blob), stream.IN_CHUNK)]
    t_d = time.time() - t0
assert b"".join(out) == data
# the library alone, called the way src/raw-file-out.c:83-123 does: the whole buffer as input, the output handed over in pieces
import ctypes as C
from zstdlite_amd import _lib
L = _lib.lib()
src = np.frombuffer(data, dtype=np.uint8)
obuf = np.empty(1 << 20, dtype=np.uint8)
sink = np.empty(L.zl_compressBound(len(data)), dtype=np.uint8)
for it in range(2):
    L.zl_CCtx_reset(cctx._ptr, 1)
    inb = stream._InBuffer(src.ctypes.data, src.size, 0); got = 0
    t0 = time.time()
    while True:
        outb = stream._OutBuffer(obuf.ctypes.data, obuf.size, 0)
        rem = L.zl_compressStream2(cctx._ptr, C.byref(outb), C.byref(inb), 2)
        assert not _lib.is_error(rem)
        sink[got:got + outb.pos] = obuf[:outb.pos]; got += outb.pos
        if rem == 0: break
    t_lib = time.time() - t0
assert sink[:got].tobytes() == blob
dst = np.empty(len(data), dtype=np.uint8)
csrc = np.frombuffer(blob, dtype=np.uint8)
for it in range(2):  # src/raw-file-in.c:87-103: the output buffer is the whole result, the input arrives in pieces
    L.zl_DCtx_reset(dctx._ptr, 1)
    outb = stream._OutBuffer(dst.ctypes.data, dst.size, 0)
    t0 = time.time()
    for o in range(0, csrc.size, 16 << 20):
        part = csrc[o:o + (16 << 20)]
        inb = stream._InBuffer(part.ctypes.data, part.size, 0)
        while inb.pos < inb.size:
            r = L.zl_decompressStream(dctx._ptr, C.byref(outb), C.byref(inb))
            assert not _lib.is_error(r)
    t_dlib = time.time() - t0
assert outb.pos == len(data) and dst.tobytes() == data
print(f"library alone: stream compress (whole buffer in, 1 MiB output pieces) {len(data) / t_lib / 1e9:.2f} GB/s, stream decompress (16 MiB input pieces, one output buffer) {len(data) / t_dlib / 1e9:.2f} GB/s")
ref = Ref()
t0 = time.time(); rblob = ref_stream_compress(ref, data, pledge=True, piece=131072); t_rc = time.time() - t0
t0 = time.time(); rout = ref_stream_decompress(ref, rblob); t_rd = time.time() - t0
assert rout == data
for it in range(2):
    t0 = time.time(); d = stream.Decompressor(dctx._ptr)
    out = [d.feed(rblob[o:o + stream.IN_CHUNK]) for o in range(0, len(rblob), stream.IN_CHUNK)]
    t_d2 = time.time() - t0
assert b"".join(out) == data
n = len(data)
print(f"{mb} MiB mix, host buffers, 64 MiB pieces: stream compress {n / t_c / 1e9:.2f} GB/s (reference {n / t_rc / 1e9:.2f}), stream decompress of our stream {n / t_d / 1e9:.2f} GB/s, "
      f"of the reference's single-frame stream {n / t_d2 / 1e9:.2f} GB/s (reference {n / t_rd / 1e9:.2f}); ratio {n / len(blob):.3f} vs {n / len(rblob):.3f}")
