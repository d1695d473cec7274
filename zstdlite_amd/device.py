"""Device-resident front end: torch uint8 CUDA tensors in, torch tensors out, through the
device entry points of the C-ABI (zl_compress_device_* / zl_decompress_device_*).  torch is
used only for device memory and streams; all work happens in libzl_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import FrameT, ZstdError, is_error

CHUNK = 128 << 10
ZL_c_compressionLevel, ZL_c_checksumFlag, ZL_c_profile, ZL_d_profile, ZL_d_forceIgnoreChecksum = 100, 201, 9001, 9002, 1002
ZL_c_frameLayout = 9003


class DeviceCodec:
    """One compression + one decompression context bound to the current torch device."""

    def __init__(self, level=3, include_checksum=False, validate_checksum=True, profile=False, frame_layout="auto"):
        if not torch.cuda.is_available():
            raise ZstdError("DeviceCodec needs a HIP device; there is no CPU fallback")
        self.L = _lib.lib()
        self.cctx = self.L.zl_createCCtx()
        self.dctx = self.L.zl_createDCtx()
        self.L.zl_CCtx_setParameter(self.cctx, ZL_c_compressionLevel, level)
        if include_checksum:
            self.L.zl_CCtx_setParameter(self.cctx, ZL_c_checksumFlag, 1)
        # device entries default to independent frames; "one_frame" = the windowed single-frame layout (include/zl_hip.h)
        self.L.zl_CCtx_setParameter(self.cctx, ZL_c_frameLayout, {"auto": 0, "frames": 1, "one_frame": 2}[frame_layout])
        if not validate_checksum:
            self.L.zl_DCtx_setParameter(self.dctx, ZL_d_forceIgnoreChecksum, 1)
        if profile:  # True / 1: per-kernel HIP events; 2: also in-kernel phase stamps (slows the kernels down)
            self.L.zl_CCtx_setParameter(self.cctx, ZL_c_profile, int(profile))
            self.L.zl_DCtx_setParameter(self.dctx, ZL_d_profile, int(profile))

    def close(self):
        if self.cctx:
            self.L.zl_freeCCtx(self.cctx)
            self.L.zl_freeDCtx(self.dctx)
            self.cctx = self.dctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        """The stream the library is to enqueue on: torch's current stream -- unless that is the default stream, whose handle is
        NULL, which the C-ABI reads as "the context's own stream" (include/zl_hip.h): a non-blocking stream that is NOT ordered
        behind what torch has queued on the default stream.  Torch's work is drained first in that case, so that a tensor that
        is still being produced (torch.randint(...) a line above) is complete before the kernels read it."""
        cur = torch.cuda.current_stream()
        if cur.cuda_stream == 0:
            cur.synchronize()
        return C.c_void_p(cur.cuda_stream)

    def bound(self, n):
        return self.L.zl_compressBound(n)

    def compress(self, src, dst=None):
        """src: uint8 CUDA tensor.  Returns (dst tensor, total compressed bytes, frame sizes)."""
        assert src.dtype == torch.uint8 and src.is_cuda and src.is_contiguous()
        n = src.numel()
        cap = self.bound(n)
        if dst is None:
            dst = torch.empty(cap, dtype=torch.uint8, device=src.device)
        assert dst.numel() >= cap
        r = self.L.zl_compress_device_async(self.cctx, C.c_void_p(dst.data_ptr()), dst.numel(), C.c_void_p(src.data_ptr()), n,
                                            self._stream())
        if is_error(r):
            raise ZstdError(f"zl_compress_device_async: {_lib.error_name(r)}")
        nframes = self.L.zl_compress_device_nframes(n)
        sizes = (C.c_size_t * nframes)()
        total = self.L.zl_compress_device_finish(self.cctx, sizes)
        if is_error(total):
            raise ZstdError(f"zl_compress_device_finish: {_lib.error_name(total)}")
        return dst, total, np.frombuffer(sizes, dtype=np.uint64).copy()

    @staticmethod
    def frame_table(frame_sizes, content_sizes):
        """Back-to-back frames with known content sizes -> zl_frame_t table."""
        fs = np.asarray(frame_sizes, dtype=np.uint64)
        cs = np.asarray(content_sizes, dtype=np.uint64)
        tbl = np.zeros((len(fs), 4), dtype=np.uint64)
        tbl[:, 0] = np.concatenate([[0], np.cumsum(fs)[:-1]])
        tbl[:, 1] = fs
        tbl[:, 2] = np.concatenate([[0], np.cumsum(cs)[:-1]])
        tbl[:, 3] = cs
        return tbl

    @staticmethod
    def chunk_content_sizes(n):
        k = max(1, -(-n // CHUNK))
        cs = np.full(k, CHUNK, dtype=np.uint64)
        cs[-1] = n - (k - 1) * CHUNK
        return cs

    def decompress(self, src, table, dst, chunk_sizes=None, host_copy=None):
        """Decode the frames listed in `table` (n x 4 uint64: src_off, src_size, dst_off, dst_cap)
        from the uint8 CUDA tensor src into dst.  Returns (total bytes, per-frame results).
        chunk_sizes: for ONE frame written by compress() in the one-frame layout, the per-chunk sizes it
        returned (zl_decompress_device_hint: the decoder then walks the frame's blocks a lane per chunk)."""
        assert src.is_cuda and dst.is_cuda
        tbl = np.ascontiguousarray(table, dtype=np.uint64)
        n = tbl.shape[0]
        if host_copy is not None:  # numpy uint8 array with the same bytes as src (zl_decompress_device_hint_host_copy)
            self.L.zl_decompress_device_hint_host_copy(self.dctx, C.c_void_p(host_copy.ctypes.data), host_copy.size)
        if chunk_sizes is not None and n == 1:
            cs = np.ascontiguousarray(chunk_sizes, dtype=np.uint64)
            self.L.zl_decompress_device_hint(self.dctx, cs.ctypes.data_as(C.POINTER(C.c_size_t)), len(cs))
        r = self.L.zl_decompress_device_async(self.dctx, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                              tbl.ctypes.data_as(C.POINTER(FrameT)), n, self._stream())
        if is_error(r):
            raise ZstdError(f"zl_decompress_device_async: {_lib.error_name(r)}")
        res = (C.c_size_t * max(n, 1))()
        total = self.L.zl_decompress_device_finish(self.dctx, res)
        if is_error(total):
            raise ZstdError(f"zl_decompress_device_finish: {_lib.error_name(total)}")
        return total, np.frombuffer(res, dtype=np.uint64)[:n].copy()

    def kernel_ms(self):
        """{'match','entropy','scan','gather','decode'} device milliseconds of the last calls."""
        c = (C.c_double * 4)()
        d = (C.c_double * 5)()
        out = {}
        if not is_error(self.L.zl_cctx_kernel_ms(self.cctx, c)):
            out.update(match=c[0], entropy=c[1], scan=c[2], gather=c[3])
            f = C.c_double(0)
            if not is_error(self.L.zl_cctx_far_ms(self.cctx, C.byref(f))) and f.value:
                out.update(far=f.value)
        if not is_error(self.L.zl_dctx_kernel_ms(self.dctx, d)):
            out.update(decode=d[0], d_index=d[1], d_entropy=d[2], d_exec=d[3], d_general=d[4])
        return out


# ------------------------------------------------------------------------------------------
# synthetic inputs on the device; same distributions as
# tests/zl_testlib.py, generated with torch so that 1 GiB takes seconds, not minutes
# ------------------------------------------------------------------------------------------
_VOCAB = [b"the", b"of", b"and", b"compression", b"zstd", b"frame", b"block", b"literal", b"sequence", b"offset",
          b"match", b"length", b"huffman", b"entropy", b"table", b"state", b"window", b"a", b"to", b"in", b"is",
          b"that", b"for", b"with", b"as", b"raw", b"vector", b"serialize", b"stream", b"dictionary", b"checksum",
          b"wavefront"]


def gen_device(kind, n, device, seed=0x9E3779B9):
    g = torch.Generator(device=device)
    g.manual_seed(seed + {"rand": 0, "iid10": 1, "int20": 2, "text": 3}[kind])
    if kind == "rand":
        return torch.randint(0, 256, (n,), dtype=torch.uint8, device=device, generator=g)
    if kind == "iid10":
        return torch.randint(1, 11, (n,), dtype=torch.uint8, device=device, generator=g)
    if kind == "int20":
        k = (n + 3) // 4
        return torch.randint(0, 1 << 20, (k,), dtype=torch.int32, device=device, generator=g).view(torch.uint8)[:n].contiguous()
    if kind == "text":
        lens = torch.tensor([len(w) for w in _VOCAB], device=device)
        flat = torch.tensor(list(b"".join(_VOCAB)), dtype=torch.uint8, device=device)
        starts = torch.cumsum(lens, 0) - lens
        out = torch.empty(n, dtype=torch.uint8, device=device)
        pos = 0
        while pos < n:
            k = 1 << 21
            idx = torch.randint(0, 32, (k,), device=device, generator=g)
            sep = torch.where(torch.randint(0, 8, (k,), device=device, generator=g) == 0, 10, 32).to(torch.uint8)
            wl = lens[idx] + 1
            ends = torch.cumsum(wl, 0)
            total = int(ends[-1])
            word_of = torch.repeat_interleave(torch.arange(k, device=device), wl)
            pos_in = torch.arange(total, device=device) - (ends - wl)[word_of]
            is_sep = pos_in == (wl[word_of] - 1)
            src_idx = starts[idx][word_of] + torch.minimum(pos_in, lens[idx][word_of] - 1)
            chunk = torch.where(is_sep, sep[word_of], flat[src_idx])
            take = min(total, n - pos)
            out[pos:pos + take] = chunk[:take]
            pos += take
        return out
    raise ValueError(kind)


def gen_mixed(n, device, shard=0):
    """The bench workload: n bytes, one quarter each of rand / iid10 / text / int20 (the four
    inputs of BASELINE.md section 2, which quotes the reference's ratio on 256 MiB of each).
    shard: which n-byte shard of a larger vector this is (bench.py --gpus N: rank r holds shard r; every shard has
    its own generator seeds, shard 0 is the one-GPU vector)."""
    q = n // 4
    parts = [gen_device(k, q if i < 3 else n - 3 * q, device, seed=0x9E3779B9 + 0x01000193 * shard) for i, k in enumerate(("rand", "iid10", "text", "int20"))]
    return torch.cat(parts), ("rand", "iid10", "text", "int20"), q
