"""Which pairs of streams give the link both directions at once?  One 1 GiB upload on stream i, one 1 GiB download on stream j."""
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
streams = []
for _ in range(12):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); streams.append(s)
def up(s): hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, G, 1, streams[s]); hip.hipStreamSynchronize(streams[s])
def down(s): hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), G, 2, streams[s]); hip.hipStreamSynchronize(streams[s])
def timed(fns):
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for i in (0, 1, 2):
    row = []
    for j in range(12):
        if j == i: row.append("  -  "); continue
        row.append(f"{timed([lambda: up(i), lambda: down(j)]):5.1f}")
    print(f"up on stream {i}, down on stream j:", " ".join(row))
# the same from ONE host thread (both copies queued, then both streams waited for)
def one_thread(i, j):
    hip.hipMemcpyAsync(d_a.data_ptr(), h_a.ctypes.data, G, 1, streams[i]); hip.hipMemcpyAsync(h_b.ctypes.data, d_b.data_ptr(), G, 2, streams[j])
    hip.hipStreamSynchronize(streams[i]); hip.hipStreamSynchronize(streams[j])
print("one host thread:", " ".join(f"{timed([lambda: one_thread(0, j)]):5.1f}" for j in range(1, 8)))
