"""Does earlier traffic on the null stream change which streams get both directions of the link at once?"""
import ctypes as C, threading, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
G = 1 << 30
d_a = torch.empty(G, dtype=torch.uint8, device=dev); d_b = torch.empty(G, dtype=torch.uint8, device=dev)
h_a = np.random.randint(0, 255, G, dtype=np.uint8); h_b = np.zeros(G, dtype=np.uint8)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode == "null_pageable":   # synchronous copies of pageable memory first, as scripts/dev/exp_duplex.py did
    hip.hipMemcpy(d_a.data_ptr(), h_a.ctypes.data, G, 1); hip.hipMemcpy(h_b.ctypes.data, d_b.data_ptr(), G, 2)
hip.hipHostRegister(h_a.ctypes.data, G, 0); hip.hipHostRegister(h_b.ctypes.data, G, 0)
if mode == "null_pinned":
    hip.hipMemcpy(d_a.data_ptr(), h_a.ctypes.data, G, 1); hip.hipMemcpy(h_b.ctypes.data, d_b.data_ptr(), G, 2)
S = {}
for name in "ab":
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
print(mode, f"alone: up {timed([lambda: up('a')]):.1f} down {timed([lambda: down('b')]):.1f}; together {timed([lambda: up('a'), lambda: down('b')]):.1f} ms;",
      f"partial sizes (660 MiB up, 1 GiB down): {timed([lambda: up('a', 660 << 20), lambda: down('b')]):.1f} ms")
