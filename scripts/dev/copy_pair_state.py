"""Does a fresh pair of streams (upload on one, download on the other) get two copy engines -- at several moments in the life of
a process that uses torch and this library?  (what zl_bind_copy_pair depends on; scripts/dev/src/copy_engines.cpp is the same
question in a bare HIP process)"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamCreateWithPriority.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
B = 32 << 20
H2D, D2H = 1, 2


def pair_test(what, prio=None, first=None, delay=0.0):
    hp, dp = C.c_void_p(), C.c_void_p()
    assert hip.hipHostMalloc(C.byref(hp), 2 * B, 0) == 0 and hip.hipMalloc(C.byref(dp), 2 * B) == 0
    a, b = C.c_void_p(), C.c_void_p()
    if prio is None:
        hip.hipStreamCreateWithFlags(C.byref(a), 1); hip.hipStreamCreateWithFlags(C.byref(b), 1)
    else:
        hip.hipStreamCreateWithPriority(C.byref(a), 1, prio[0]); hip.hipStreamCreateWithPriority(C.byref(b), 1, prio[1])
    if first == "overlap":  # b's first copy is issued while a's first copy is certainly under way
        hip.hipMemcpyAsync(dp, hp, 2 * B, H2D, a)
        time.sleep(delay)
        hip.hipMemcpyAsync(hp.value + B, dp.value + B, B, D2H, b)
        hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)
    elif first == "overlap_rev":
        hip.hipMemcpyAsync(hp.value + B, dp.value + B, B, D2H, b)
        time.sleep(delay)
        hip.hipMemcpyAsync(dp, hp, B, H2D, a)
        hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)
    best = [1e9, 1e9, 1e9]
    for _ in range(3):
        t0 = time.perf_counter()
        hip.hipMemcpyAsync(dp, hp, B, H2D, a); hip.hipStreamSynchronize(a)
        t1 = time.perf_counter()
        hip.hipMemcpyAsync(hp.value + B, dp.value + B, B, D2H, b); hip.hipStreamSynchronize(b)
        t2 = time.perf_counter()
        hip.hipMemcpyAsync(dp, hp, B, H2D, a); hip.hipMemcpyAsync(hp.value + B, dp.value + B, B, D2H, b)
        hip.hipStreamSynchronize(a); hip.hipStreamSynchronize(b)
        t3 = time.perf_counter()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1), min(best[2], t3 - t2)]
    print(f"{what:60s} up {best[0] * 1e3:.2f} dn {best[1] * 1e3:.2f} both {best[2] * 1e3:.2f} ms -> {'TWO ENGINES' if best[2] < 0.75 * (best[0] + best[1]) else 'taking turns'}", flush=True)
    hip.hipStreamDestroy(a); hip.hipStreamDestroy(b)
    hip.hipFree(dp); hip.hipHostFree(hp)


torch.cuda.init()
print([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][::8])
pair_test("after torch.cuda.init()")
for d in (0.0, 0.0001, 0.0003, 0.001):
    pair_test(f"first copies overlapping, second {d * 1e6:.0f} us behind", first="overlap", delay=d)
    pair_test(f"first copies overlapping (download first), {d * 1e6:.0f} us", first="overlap_rev", delay=d)
x = torch.ones(1 << 20, device="cuda")
pair_test("after a tensor on the device")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
y = x.cpu()
pair_test("after torch's first download")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
n = 256 << 20
host = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy()
pair_test("after gen_mixed + .cpu()")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
cctx = L.zl_createCCtx()
r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
pair_test("after zl_compress2 (cctx alive)")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
pair_test("  ... low / high priority pair", (1, -1))
L.zl_freeCCtx(cctx)
pair_test("after zl_freeCCtx")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
dctx = L.zl_createDCtx()
L.zl_DCtx_setParameter(dctx, 9005, 0)
assert L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r) == n
pair_test("after zl_decompressDCtx without the pipeline (dctx alive)")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
pair_test("  ... low / high priority pair", (1, -1))
L.zl_freeDCtx(dctx)
pair_test("after zl_freeDCtx")
pair_test("  ... first copies overlapping", first="overlap", delay=0.0001)
pair_test("  ... low / high priority pair", (1, -1))
