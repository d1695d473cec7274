"""Throughput of the streaming entries on host buffers, library alone: zl_compressStream2 fed the bench vector
in 8 MiB pieces into a preallocated output (pledged = one frame, and unpledged), zl_decompressStream fed the result in 8 MiB
pieces; against the reference's own streaming calls on one host core.  usage: python scripts/stream_bench.py [MiB=1024]
(DESIGN.md section 4 quotes these numbers; rounds 1-3 cited a script that was never committed)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import zstdlite_amd as z
from zstdlite_amd import _lib, stream
from zstdlite_amd.device import gen_mixed


def stream_compress(L, cptr, src, dst, piece, pledged):
    L.zl_CCtx_reset(cptr, 1)
    if pledged:
        L.zl_CCtx_setPledgedSrcSize(cptr, src.size)
    outb = stream._OutBuffer(dst.ctypes.data, dst.size, 0)
    n = src.size
    for off in range(0, n, piece):
        ln = min(piece, n - off)
        inb = stream._InBuffer(src.ctypes.data + off, ln, 0)
        mode = 2 if off + ln == n else 0
        while True:
            rem = L.zl_compressStream2(cptr, C.byref(outb), C.byref(inb), mode)
            assert not _lib.is_error(rem), _lib.error_name(rem)
            if inb.pos == inb.size and (mode == 0 or rem == 0):
                break
    return outb.pos


def stream_decompress(L, dptr, comp, csize, dst, piece):
    L.zl_DCtx_reset(dptr, 1)
    outb = stream._OutBuffer(dst.ctypes.data, dst.size, 0)
    for off in range(0, csize, piece):
        ln = min(piece, csize - off)
        inb = stream._InBuffer(comp.ctypes.data + off, ln, 0)
        while True:
            before = (inb.pos, outb.pos)
            r = L.zl_decompressStream(dptr, C.byref(outb), C.byref(inb))
            assert not _lib.is_error(r), _lib.error_name(r)
            if inb.pos == inb.size and (r == 0 or (inb.pos, outb.pos) == before or off + ln < csize):
                break
    return outb.pos


def main(mib=1024, log=print, host=None):
    L = _lib.lib()
    if host is None:
        host = gen_mixed(mib << 20, torch.device("cuda", 0))[0].cpu().numpy()
    n = host.size
    comp = np.empty(L.zl_compressBound(n) + (1 << 20), dtype=np.uint8); comp.fill(1)
    back = np.empty(n, dtype=np.uint8); back.fill(1)
    cctx, dctx = z.zstd_cctx(), z.zstd_dctx()
    piece = 8 << 20
    out = {"bytes": n, "piece_bytes": piece}
    for label, pledged in (("pledged_one_frame", True), ("unpledged", False)):
        tc = td = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); csize = stream_compress(L, cctx._ptr, host, comp, piece, pledged); tc = min(tc, time.perf_counter() - t0)
            t0 = time.perf_counter(); got = stream_decompress(L, dctx._ptr, comp, csize, back, piece); td = min(td, time.perf_counter() - t0)
            assert got == n and np.array_equal(back, host)
        out[label] = {"stream_compress_GBps": n / tc / 1e9, "stream_decompress_GBps": n / td / 1e9, "compressed_bytes": int(csize)}
        log(f"stream {label}: compress {n / tc / 1e9:.2f} GB/s, decompress {n / td / 1e9:.2f} GB/s, {csize} bytes")
    return out


if __name__ == "__main__":
    print(json.dumps(main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)))
