"""Does a process that has already created many streams and copied on them still get two copy engines for a new pair of streams?"""
import ctypes as C, threading, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
G = 1 << 30
nbusy = int(sys.argv[1])
d_a = torch.empty(G, dtype=torch.uint8, device=dev); d_b = torch.empty(G, dtype=torch.uint8, device=dev)
h_a = np.random.randint(0, 255, G, dtype=np.uint8); h_b = np.zeros(G, dtype=np.uint8)
hip.hipHostRegister(h_a.ctypes.data, G, 0); hip.hipHostRegister(h_b.ctypes.data, G, 0)
def mk():
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); return s
busy = [mk() for _ in range(nbusy)]
for k, s in enumerate(busy):   # sequential large copies, alternating directions: what a long-lived process has behind it
    if k & 1: hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), 64 << 20, 2, s)
    else: hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, 64 << 20, 1, s)
    hip.hipStreamSynchronize(s)
a, b = mk(), mk()
hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, 16 << 20, 1, a); hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), 16 << 20, 2, b)
hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)
def up(): hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, G, 1, a); hip.hipStreamSynchronize(a)
def down(): hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), G, 2, b); hip.hipStreamSynchronize(b)
def timed(fns):
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"{nbusy} earlier streams: a fresh pair bound together: {timed([up, down]):.1f} ms")
