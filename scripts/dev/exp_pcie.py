"""Host <-> device copy rates on this box: pageable, pinned, and the cost of pinning in place."""
import time, torch, numpy as np
n = 1 << 30
dev = torch.device("cuda", 0)
d = torch.empty(n, dtype=torch.uint8, device=dev)
pag = torch.from_numpy(np.ones(n, dtype=np.uint8))
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pin.fill_(1)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
print(f"H2D pageable {n / t(lambda: d.copy_(pag)) / 1e9:.1f} GB/s   pinned {n / t(lambda: d.copy_(pin, non_blocking=True)) / 1e9:.1f} GB/s")
print(f"D2H pageable {n / t(lambda: pag.copy_(d)) / 1e9:.1f} GB/s   pinned {n / t(lambda: pin.copy_(d, non_blocking=True)) / 1e9:.1f} GB/s")
rt = torch.cuda.cudart()
buf = np.ones(n, dtype=np.uint8)
t0 = time.perf_counter(); r = rt.cudaHostRegister(buf.ctypes.data, n, 0); t1 = time.perf_counter()
print(f"hipHostRegister 1 GiB: {(t1 - t0) * 1e3:.1f} ms (rc {r})")
reg = torch.from_numpy(buf)
print(f"H2D registered {n / t(lambda: d.copy_(reg, non_blocking=True)) / 1e9:.1f} GB/s")
t0 = time.perf_counter(); rt.cudaHostUnregister(buf.ctypes.data); print(f"unregister {(time.perf_counter() - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); buf2 = buf.copy(); print(f"host memcpy 1 GiB (1 thread): {n / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
