"""Does the rate of a download depend on the DEVICE buffer (size of the allocation, offset, touched or not)?"""
import ctypes as C
import time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
B = 32 << 20
torch.cuda.init()
hp = C.c_void_p(); assert hip.hipHostMalloc(C.byref(hp), B, 0) == 0
C.memset(hp, 1, B)
s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1)


def rate(dev, what):
    tu = td = 1e9
    first = None
    for k in range(5):
        t0 = time.perf_counter(); hip.hipMemcpyAsync(hp, dev, B, 2, s); hip.hipStreamSynchronize(s); t = time.perf_counter() - t0
        first = first or t
        td = min(td, t)
        t0 = time.perf_counter(); hip.hipMemcpyAsync(dev, hp, B, 1, s); hip.hipStreamSynchronize(s); tu = min(tu, time.perf_counter() - t0)
    print(f"{what:60s} down {B / td / 1e9:5.1f} GB/s (first {B / first / 1e9:5.1f}), up {B / tu / 1e9:5.1f} GB/s", flush=True)


for size in (64 << 20, 700 << 20, 2 << 30, 6 << 30):
    dp = C.c_void_p(); assert hip.hipMalloc(C.byref(dp), size) == 0
    rate(dp.value, f"hipMalloc {size >> 20} MiB, offset 0, untouched")
    rate(dp.value + size - B, f"hipMalloc {size >> 20} MiB, last 32 MiB, untouched")
    hip.hipMemset(dp, 3, size)
    rate(dp.value + size // 2, f"hipMalloc {size >> 20} MiB, middle, after a memset")
    hip.hipFree(dp)
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
rate(x.data_ptr(), "torch tensor of 1 GiB, untouched")
x.fill_(7); torch.cuda.synchronize()
rate(x.data_ptr() + (512 << 20), "torch tensor of 1 GiB, middle, filled")
# after big frees: a new allocation reuses freed memory
dp = C.c_void_p(); assert hip.hipMalloc(C.byref(dp), 1 << 30) == 0
rate(dp.value, "hipMalloc 1 GiB after the frees, offset 0")
