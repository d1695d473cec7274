"""Hypothesis: a stream's copy engine is fixed by the direction of its FIRST copy."""
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
hip.hipHostRegister(h_a.ctypes.data, G, 0); hip.hipHostRegister(h_b.ctypes.data, G, 0)
S = {}
for name in "abcdef":
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); S[name] = s
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
# first uses (tiny copies): a, c for upload; b, d for download; e, f untouched
up("a", 4096); down("b", 4096); up("c", 4096); down("d", 4096)
for u, dn in (("a", "b"), ("c", "d"), ("a", "c"), ("b", "d"), ("b", "a"), ("d", "c"), ("e", "f"), ("f", "e"), ("a", "e"), ("e", "b")):
    print(f"up on {u}, down on {dn}: {timed([lambda: up(u), lambda: down(dn)]):5.1f} ms")
