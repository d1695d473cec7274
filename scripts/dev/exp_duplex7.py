"""Pageable host memory on two streams that were bound to two copy engines (first copies in flight together, pinned scratch)."""
import ctypes as C, threading, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
G = 1 << 30
d_a = torch.empty(G, dtype=torch.uint8, device=dev); d_b = torch.empty(G, dtype=torch.uint8, device=dev)
h_a = np.random.randint(0, 255, G, dtype=np.uint8); h_b = np.zeros(G, dtype=np.uint8)   # pageable
p_a = np.zeros(16 << 20, dtype=np.uint8); p_b = np.zeros(16 << 20, dtype=np.uint8)
hip.hipHostRegister(p_a.ctypes.data, p_a.size, 0); hip.hipHostRegister(p_b.ctypes.data, p_b.size, 0)
S = []
for _ in range(2):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); S.append(s)
# bind: both first copies in flight at once
hip.hipMemcpyAsync(d_a.data_ptr(), p_a.ctypes.data, p_a.size, 1, S[0]); hip.hipMemcpyAsync(p_b.ctypes.data, d_b.data_ptr(), p_b.size, 2, S[1])
hip.hipStreamSynchronize(S[0]); hip.hipStreamSynchronize(S[1])
def up(n=G): hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, n, 1, S[0]); hip.hipStreamSynchronize(S[0])
def down(n=G): hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), n, 2, S[1]); hip.hipStreamSynchronize(S[1])
def timed(fns):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"pageable: up {timed([up]):.1f} ms, down {timed([down]):.1f} ms, together {timed([up, down]):.1f} ms")
hip.hipHostRegister(h_a.ctypes.data, G, 0); hip.hipHostRegister(h_b.ctypes.data, G, 0)
print(f"the same buffers registered: up {timed([up]):.1f} ms, down {timed([down]):.1f} ms, together {timed([up, down]):.1f} ms")
