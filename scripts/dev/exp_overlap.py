"""Does running two half-size jobs on two streams beat running them back to back?  (decides whether
pipelining groups over streams is worth building)"""
import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
from zstdlite_amd import _lib
dev = torch.device("cuda", 0)
n = 1 << 30
src, _, _ = gen_mixed(n, dev)
# interleave so that both halves hold the same mix
src = src.view(4, 2, -1).transpose(0, 1).contiguous().view(-1)
h = n // 2
cod = [DeviceCodec(), DeviceCodec()]
L = cod[0].L
comp = [torch.empty(cod[0].bound(h), dtype=torch.uint8, device=dev) for _ in range(2)]
back = [torch.empty(h, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run_c(two_streams):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2):
        st = streams[i] if two_streams else streams[0]
        r = L.zl_compress_device_async(cod[i].cctx, C.c_void_p(comp[i].data_ptr()), comp[i].numel(), C.c_void_p(src[i*h:].data_ptr()), h, C.c_void_p(st.cuda_stream))
        assert not _lib.is_error(r)
    tot = []
    sizes = []
    for i in range(2):
        nf = L.zl_compress_device_nframes(h); sz = (C.c_size_t * nf)()
        tot.append(L.zl_compress_device_finish(cod[i].cctx, sz)); sizes.append(np.frombuffer(sz, dtype=np.uint64).copy())
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, tot, sizes
def run_d(two_streams, sizes):
    tbls = [cod[0].frame_table(sizes[i], cod[0].chunk_content_sizes(h)) for i in range(2)]
    tb = [np.ascontiguousarray(t, dtype=np.uint64) for t in tbls]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2):
        st = streams[i] if two_streams else streams[0]
        r = L.zl_decompress_device_async(cod[i].dctx, C.c_void_p(back[i].data_ptr()), C.c_void_p(comp[i].data_ptr()), tb[i].ctypes.data_as(C.POINTER(_lib.FrameT)), tb[i].shape[0], C.c_void_p(st.cuda_stream))
        assert not _lib.is_error(r)
    for i in range(2):
        res = (C.c_size_t * tb[i].shape[0])()
        assert not _lib.is_error(L.zl_decompress_device_finish(cod[i].dctx, res))
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for rep in range(3):
    for two in (False, True):
        ms, tot, sizes = run_c(two)
        md = run_d(two, sizes)
        print(f"two_streams={two}: compress {ms:.2f} ms  decompress {md:.2f} ms")
assert torch.equal(back[0], src[:h]) and torch.equal(back[1], src[h:])
