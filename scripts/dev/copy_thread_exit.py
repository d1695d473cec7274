"""Does a host thread that issued copies on a stream and then EXITED leave a copy engine unusable for later streams?"""
import ctypes as C
import threading
import time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
B = 16 << 20
torch.cuda.init()
hp, dp = C.c_void_p(), C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), 4 * B, 0) == 0 and hip.hipMalloc(C.byref(dp), 4 * B) == 0


def stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s


def t_of(fn):
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3


def up(s): hip.hipMemcpyAsync(dp, hp, B, 1, s); hip.hipStreamSynchronize(s)
def down(s): hip.hipMemcpyAsync(hp.value + 2 * B, dp.value + 2 * B, B, 2, s); hip.hipStreamSynchronize(s)
def both(a, b): hip.hipMemcpyAsync(dp, hp, B, 1, a); hip.hipMemcpyAsync(hp.value + 2 * B, dp.value + 2 * B, B, 2, b); hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)


def fresh_pair(what):
    a, b = stream(), stream()
    both(a, b)
    tu = min(t_of(lambda: up(a)) for _ in range(3)); td = min(t_of(lambda: down(b)) for _ in range(3)); tb = min(t_of(lambda: both(a, b)) for _ in range(3))
    print(f"{what:64s} up {tu:.2f} down {td:.2f} both {tb:.2f}", flush=True)
    hip.hipStreamDestroy(a); hip.hipStreamDestroy(b)


fresh_pair("start")
a, b = stream(), stream()
th = threading.Thread(target=lambda: [down(b) for _ in range(8)]); th.start()
[up(a) for _ in range(8)]
th.join()
fresh_pair("after a thread downloaded on b beside uploads on a, and exited")
hip.hipStreamDestroy(a); hip.hipStreamDestroy(b)
fresh_pair("after those two streams were destroyed")
# pageable traffic from two threads, as the library's pipeline does
import numpy as np
big = np.empty(256 << 20, dtype=np.uint8); big.fill(1)
dbig = C.c_void_p(); assert hip.hipMalloc(C.byref(dbig), 256 << 20) == 0
a, b = stream(), stream()
both(a, b)


def pageable_down():
    for k in range(8):
        hip.hipMemcpyAsync(big.ctypes.data + k * (32 << 20), dbig.value + k * (32 << 20), 32 << 20, 2, b); hip.hipStreamSynchronize(b)


th = threading.Thread(target=pageable_down); th.start()
for k in range(8):
    hip.hipMemcpyAsync(dbig.value + k * (32 << 20), big.ctypes.data + k * (32 << 20), 32 << 20, 1, a); hip.hipStreamSynchronize(a)
th.join()
fresh_pair("after PAGEABLE traffic from two threads (streams alive)")
hip.hipStreamDestroy(a); hip.hipStreamDestroy(b)
fresh_pair("... streams destroyed")
hip.hipFree(dbig)
fresh_pair("... device buffer freed")
e = C.c_void_p(); hip.hipHostMalloc(C.byref(e), 64 << 20, 0); hip.hipHostFree(e)
fresh_pair("after a pinned buffer came and went")
