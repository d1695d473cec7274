"""What makes two streams share the link in both directions?  argv: number of streams created, warm-up bytes (0 = none), order"""
import ctypes as C, threading, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
G = 1 << 30
nstreams, warm, order = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
d_a = torch.empty(G, dtype=torch.uint8, device=dev); d_b = torch.empty(G, dtype=torch.uint8, device=dev)
h_a = np.random.randint(0, 255, G, dtype=np.uint8); h_b = np.zeros(G, dtype=np.uint8)
hip.hipHostRegister(h_a.ctypes.data, G, 0); hip.hipHostRegister(h_b.ctypes.data, G, 0)
S = []
for _ in range(nstreams):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); S.append(s)
def up(s, n=G): hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, n, 1, S[s]); hip.hipStreamSynchronize(S[s])
def down(s, n=G): hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), n, 2, S[s]); hip.hipStreamSynchronize(S[s])
def timed(fns):
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
if warm:
    if order == "ud": up(0, warm); down(1, warm)
    else: down(1, warm); up(0, warm)
t1 = timed([lambda: up(0), lambda: down(1)])
t2 = timed([lambda: up(0), lambda: down(1)])
print(f"streams {nstreams} warm-up {warm} order {order}: together {t1:.1f} ms, again {t2:.1f} ms")
