"""Host-side mirror of the reference's streaming glue over the C-ABI's streaming twins
zl_compressStream2 / zl_decompressStream (include/zl_hip.h):

  write_file / read_file   src/raw-file-out.c:31-125, src/raw-file-in.c:32-125 (and the serialize-file
                           variants): what use_file_streaming=TRUE selects in zstd_compress(),
                           zstd_decompress(), zstd_serialize(), zstd_unserialize()
  zstdfile                 src/zstdfile.c (R/zstdfile.R): a file connection whose bytes are zstd on disk;
                           write() / read() / readlines() / close(), text or binary

Buffers are the reference's sizes scaled up: the glue hands 128 KiB at a time to libzstd
(ZSTD_CStreamInSize); the kernels want whole jobs, so 64 MiB is read and written per call here
(any size is correct, the library cuts jobs itself)."""
import ctypes as C
import io

import numpy as np

from . import _lib
from ._lib import is_error

IN_CHUNK = 64 << 20
OUT_CHUNK = (64 << 20) + (1 << 16)


class _InBuffer(C.Structure):
    _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


class _OutBuffer(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


E_CONTINUE, E_FLUSH, E_END = 0, 1, 2
RESET_SESSION = 1


class ZstdStreamError(RuntimeError):
    pass


def _check(r, who):
    if is_error(r):
        raise ZstdStreamError(f"{who}: {_lib.error_name(r)}")
    return r


class Compressor:
    """One streaming compression session after the other on a cctx (ZSTD_compressStream2 call pattern
    of src/raw-file-out.c:83-123): feed() returns the compressed bytes that became available."""

    def __init__(self, cctx_ptr, pledged=None):
        self.L = _lib.lib()
        self.c = cctx_ptr
        self.out = np.empty(OUT_CHUNK, dtype=np.uint8)
        _check(self.L.zl_CCtx_reset(self.c, RESET_SESSION), "zl_CCtx_reset")
        if pledged is not None:
            _check(self.L.zl_CCtx_setPledgedSrcSize(self.c, pledged), "zl_CCtx_setPledgedSrcSize")

    def _pump(self, data, mode):
        src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.empty(0, dtype=np.uint8)
        inb = _InBuffer(src.ctypes.data if len(data) else None, len(data), 0)
        pieces = []
        while True:
            outb = _OutBuffer(self.out.ctypes.data, self.out.size, 0)
            rem = _check(self.L.zl_compressStream2(self.c, C.byref(outb), C.byref(inb), mode), "zl_compressStream2")
            if outb.pos:
                pieces.append(self.out[:outb.pos].tobytes())
            if inb.pos == inb.size and (rem == 0 or (mode == E_CONTINUE and outb.pos == 0)):
                break
        return b"".join(pieces)

    def feed(self, data):
        return self._pump(data, E_CONTINUE)

    def flush(self):
        return self._pump(b"", E_FLUSH)

    def finish(self, data=b""):
        return self._pump(data, E_END)


class Decompressor:
    """ZSTD_decompressStream call pattern of src/raw-file-in.c:94-103: feed() returns the bytes that became available."""

    def __init__(self, dctx_ptr):
        self.L = _lib.lib()
        self.d = dctx_ptr
        self.out = np.empty(OUT_CHUNK, dtype=np.uint8)
        self.hint = 0
        _check(self.L.zl_DCtx_reset(self.d, RESET_SESSION), "zl_DCtx_reset")

    def feed(self, data):
        src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.empty(0, dtype=np.uint8)
        inb = _InBuffer(src.ctypes.data if len(data) else None, len(data), 0)
        pieces = []
        while True:
            outb = _OutBuffer(self.out.ctypes.data, self.out.size, 0)
            self.hint = _check(self.L.zl_decompressStream(self.d, C.byref(outb), C.byref(inb)), "zl_decompressStream")
            if outb.pos:
                pieces.append(self.out[:outb.pos].tobytes())
            if inb.pos == inb.size and outb.pos < outb.size:
                break
        return b"".join(pieces)

    @property
    def at_frame_boundary(self):
        return self.hint == 0


def write_file(path, data, cctx_ptr, who):
    """src/raw-file-out.c:31-125: pledge the size, stream the buffer out in pieces, end the frame."""
    try:
        f = open(path, "wb")
    except OSError:
        raise ZstdStreamError(f"{who}(): Couldn't open file for output '{path}'")
    with f:
        comp = Compressor(cctx_ptr, pledged=len(data))
        view = memoryview(data)
        for off in range(0, len(view), IN_CHUNK):
            f.write(comp.feed(view[off:off + IN_CHUNK]))
        f.write(comp.finish())


def read_file(path, dctx_ptr, who):
    """src/raw-file-in.c:32-125: stream the file through the decoder.  (The reference sizes its result from
    the first frame's header; this stream is many frames, so the pieces are collected instead.)"""
    try:
        f = open(path, "rb")
    except OSError:
        raise ZstdStreamError(f"{who}(): Couldn't open file '{path}'")
    with f:
        dec = Decompressor(dctx_ptr)
        pieces = []
        while True:
            chunk = f.read(IN_CHUNK)
            if not chunk:
                break
            pieces.append(dec.feed(chunk))
        if not dec.at_frame_boundary:
            raise ZstdStreamError(f"{who}(): file '{path}' ends inside a frame")
        return b"".join(pieces)


class zstdfile(io.RawIOBase):
    """R/zstdfile.R -> src/zstdfile.c: a file whose content is zstd on disk.  Modes 'rb' / 'wb' (binary)
    and 'r' / 'w' (text, UTF-8): zstdfile(path, 'w').write(...) ... close(); zstdfile(path).read()."""

    def __init__(self, description, open="rb", *, cctx=None, dctx=None, **opts):
        super().__init__()
        from . import _init_cctx_with_opts, _init_dctx_with_opts
        if open not in ("r", "rb", "rt", "w", "wb", "wt"):
            raise ValueError(f"zstdfile(): mode '{open}' is not supported")  # src/zstdfile.c:152-170 (no append, no '+')
        self.text = "b" not in open
        self.writing = open[0] == "w"
        self.name = description
        if self.writing:
            self._ctx = cctx if cctx is not None else _init_cctx_with_opts(opts)
            self._f = io.open(description, "wb")
            self._comp = Compressor(self._ctx._ptr)
            self._pending = bytearray()
        else:
            self._ctx = dctx if dctx is not None else _init_dctx_with_opts(opts)
            self._f = io.open(description, "rb")
            self._dec = Decompressor(self._ctx._ptr)
            self._buf = bytearray()
            self._eof = False

    def readable(self):
        return not self.writing

    def writable(self):
        return self.writing

    def write(self, data):  # src/zstdfile.c:447-520
        if not self.writing:
            raise io.UnsupportedOperation("not writable")
        if isinstance(data, str):
            data = data.encode()
        self._pending += data
        if len(self._pending) >= IN_CHUNK:
            self._f.write(self._comp.feed(bytes(self._pending)))
            self._pending.clear()
        return len(data)

    def writelines(self, lines):
        for ln in lines:
            self.write(ln if isinstance(ln, (bytes, bytearray)) or ln.endswith("\n") else ln + "\n")

    def _fill(self, want):
        while not self._eof and (want < 0 or len(self._buf) < want):
            chunk = self._f.read(IN_CHUNK)
            if not chunk:
                self._eof = True
                if not self._dec.at_frame_boundary:
                    raise ZstdStreamError(f"zstdfile(): '{self.name}' ends inside a frame")
                break
            self._buf += self._dec.feed(chunk)

    def read(self, n=-1):  # src/zstdfile.c:317-400
        if self.writing:
            raise io.UnsupportedOperation("not readable")
        self._fill(n)
        k = len(self._buf) if n is None or n < 0 else min(n, len(self._buf))
        out = bytes(self._buf[:k])
        del self._buf[:k]
        return out.decode() if self.text else out

    def readlines(self, hint=-1):
        data = self.read()
        return data.splitlines() if self.text else data.splitlines(keepends=True)

    def close(self):  # src/zstdfile.c:217-260: the frame is ended when the connection is closed
        if self.closed:
            return
        try:
            if self.writing:
                self._f.write(self._comp.finish(bytes(self._pending)))
                self._pending.clear()
            self._f.close()
        finally:
            super().close()
