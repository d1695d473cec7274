"""How does the HIP runtime this process uses (PyTorch's) give copy engines to streams?  Sequences of downloads on one stream,
alone and beside uploads on another stream issued from a second thread, with the time of every copy.  (zl_copy_pair.h)"""
import ctypes as C
import sys
import threading
import time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
B = 32 << 20
H2D, D2H = 1, 2
torch.cuda.init()
hp, dp = C.c_void_p(), C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), 4 * B, 0) == 0 and hip.hipMalloc(C.byref(dp), 4 * B) == 0


def stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    return s


def down(s, k=0):
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(hp.value + (2 + k) * B, dp.value + (2 + k) * B, B, D2H, s); hip.hipStreamSynchronize(s)
    return (time.perf_counter() - t0) * 1e3


def up(s):
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(dp, hp, B, H2D, s); hip.hipStreamSynchronize(s)
    return (time.perf_counter() - t0) * 1e3


def fmt(ts):
    return " ".join(f"{t:.2f}" for t in ts)


a, b = stream(), stream()
print("T1 downloads alone on a fresh stream b (ms each):", fmt([down(b) for _ in range(24)]))
print("T1 uploads alone on a fresh stream a:            ", fmt([up(a) for _ in range(24)]))
stop = False


def uploader(s):
    while not stop:
        up(s)


th = threading.Thread(target=uploader, args=(a,)); th.start()
time.sleep(0.01)
print("T2 downloads on b beside uploads on a (2nd thread):", fmt([down(b) for _ in range(40)]))
c = stream()
print("T2 downloads on a NEW stream c beside the uploads: ", fmt([down(c) for _ in range(40)]))
stop = True; th.join()
print("T2 downloads on c, uploads stopped:                ", fmt([down(c) for _ in range(12)]))
print("T2 downloads on b, uploads stopped:                ", fmt([down(b) for _ in range(12)]))
# streams that come and go
for k in range(12):
    s = stream(); up(s); down(s); hip.hipStreamDestroy(s)
d, e = stream(), stream()
stop = False
th = threading.Thread(target=uploader, args=(d,)); th.start()
time.sleep(0.01)
print("T3 after 12 streams came and went: downloads on new e beside uploads on new d:", fmt([down(e) for _ in range(40)]))
stop = True; th.join()
print("T3 downloads on e, uploads stopped:", fmt([down(e) for _ in range(12)]))
res = []
th = threading.Thread(target=lambda: res.extend(down(e) for _ in range(12))); th.start(); th.join()
print("T4 downloads on e from a new thread:", fmt(res))
stop = False
th = threading.Thread(target=uploader, args=(d,)); th.start()
time.sleep(0.01)
res = []
t2 = threading.Thread(target=lambda: res.extend(down(e) for _ in range(24))); t2.start(); t2.join()
stop = True; th.join()
print("T4 downloads on e from a new thread beside uploads on d:", fmt(res))
