"""PCIe both ways at once, both size mixes (decompress: 0.66 GB up, 1 GiB down; compress: 1 GiB up, 0.66 GB down), pinned host memory,
hipMemcpyAsync on two non-blocking streams from two host threads; also 4 concurrent pieces per direction."""
import ctypes as C, threading, time
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
G = 1 << 30
d_a = torch.empty(G, dtype=torch.uint8, device=dev); d_b = torch.empty(G, dtype=torch.uint8, device=dev)
h_a = np.random.randint(0, 255, G, dtype=np.uint8); h_b = np.zeros(G, dtype=np.uint8)
print("register", hip.hipHostRegister(h_a.ctypes.data, G, 0), hip.hipHostRegister(h_b.ctypes.data, G, 0))
streams = []
for _ in range(8):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); streams.append(s)
def up(n, s=0, off=0): hip.hipMemcpyAsync(d_a.data_ptr() + off, h_a.ctypes.data + off, n, 1, streams[s]); hip.hipStreamSynchronize(streams[s])
def down(n, s=1, off=0): hip.hipMemcpyAsync(h_b.ctypes.data + off, d_b.data_ptr() + off, n, 2, streams[s]); hip.hipStreamSynchronize(streams[s])
def timed(fns):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
small = 660 << 20
for label, nu, nd in (("decompress mix", small, G), ("compress mix", G, small), ("equal", G, G)):
    a, b, c = timed([lambda: up(nu)]), timed([lambda: down(nd)]), timed([lambda: up(nu), lambda: down(nd)])
    print(f"{label}: up {a:.1f} ms, down {b:.1f} ms, both at once {c:.1f} ms (sum {a + b:.1f}) -> {(nu + nd) / c / 1e6:.1f} GB/s aggregate")
# four pieces per direction on four streams each
q = G // 4
c4 = timed([(lambda i=i: up(q, i, i * q)) for i in range(4)] + [(lambda i=i: down(q, 4 + i, i * q)) for i in range(4)])
print(f"equal, 4 + 4 streams: {c4:.1f} ms -> {2 * G / c4 / 1e6:.1f} GB/s aggregate")
for _ in range(8):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); streams.append(s)
u4 = timed([(lambda i=i: up(q, i, i * q)) for i in range(4)])
d4 = timed([(lambda i=i: down(q, 4 + i, i * q)) for i in range(4)])
print(f"one direction on 4 streams: up {G / u4 / 1e6:.1f} GB/s, down {G / d4 / 1e6:.1f} GB/s")
e = G // 8
c8 = timed([(lambda i=i: up(e, i, i * e)) for i in range(8)] + [(lambda i=i: down(e, 8 + i, i * e)) for i in range(8)])
print(f"equal, 8 + 8 streams: {c8:.1f} ms -> {2 * G / c8 / 1e6:.1f} GB/s aggregate")
c2 = timed([(lambda i=i: up(G // 2, i, i * (G // 2))) for i in range(2)] + [(lambda i=i: down(G // 2, 4 + i, i * (G // 2))) for i in range(2)])
print(f"equal, 2 + 2 streams: {c2:.1f} ms -> {2 * G / c2 / 1e6:.1f} GB/s aggregate")
# a kernel-side copy (device reads mapped host memory) beside an engine download
dp = C.c_void_p()
hip.hipHostGetDevicePointer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipHostGetDevicePointer(C.byref(dp), h_a.ctypes.data, 0)
mapped = None
try:
    import torch.utils.dlpack
except Exception:
    pass
