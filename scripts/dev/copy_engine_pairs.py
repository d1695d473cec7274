"""Pairs of streams bound with overlapping first copies, created and destroyed in turn: do later pairs get worse engines?
(the runtime PyTorch brings; zl_copy_pair.h)"""
import ctypes as C
import time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
B = 16 << 20
torch.cuda.init()
hp, dp = C.c_void_p(), C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), 4 * B, 0) == 0 and hip.hipMalloc(C.byref(dp), 4 * B) == 0


def stream():
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    return s


def t_of(fn):
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3


def up(s): hip.hipMemcpyAsync(dp, hp, B, 1, s); hip.hipStreamSynchronize(s)
def down(s): hip.hipMemcpyAsync(hp.value + 2 * B, dp.value + 2 * B, B, 2, s); hip.hipStreamSynchronize(s)
def both(a, b): hip.hipMemcpyAsync(dp, hp, B, 1, a); hip.hipMemcpyAsync(hp.value + 2 * B, dp.value + 2 * B, B, 2, b); hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)


def bound_pair():
    a, b = stream(), stream()
    both(a, b)
    return a, b


def report(what, a, b):
    tu = min(t_of(lambda: up(a)) for _ in range(3)); td = min(t_of(lambda: down(b)) for _ in range(3)); tb = min(t_of(lambda: both(a, b)) for _ in range(3))
    print(f"{what:50s} up {tu:.2f} down {td:.2f} both {tb:.2f}", flush=True)


keep = []
for k in range(6):
    a, b = bound_pair()
    report(f"pair {k} (earlier pairs destroyed)", a, b)
    hip.hipStreamDestroy(a); hip.hipStreamDestroy(b)
for k in range(6):
    a, b = bound_pair()
    report(f"pair {k} (earlier pairs ALIVE)", a, b)
    keep.append((a, b))
report("the first of the living pairs again", *keep[0])
report("the last of the living pairs again", *keep[-1])
# many copies on one pair: does its engine change with time?
a, b = keep[0]
ts = []
for k in range(200):
    ts.append(t_of(lambda: both(a, b)))
print("200 x both on the first living pair:", " ".join(f"{t:.2f}" for t in ts[::8]))
for (x, y) in keep[1:]:
    hip.hipStreamDestroy(x); hip.hipStreamDestroy(y)
ts = [t_of(lambda: both(a, b)) for _ in range(100)]
print("after the others were destroyed:     ", " ".join(f"{t:.2f}" for t in ts[::4]))
ts = [t_of(lambda: down(b)) for _ in range(40)]
print("downloads alone:                     ", " ".join(f"{t:.2f}" for t in ts[::2]))
