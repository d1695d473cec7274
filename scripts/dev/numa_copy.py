"""Does the rate of a download depend on the NUMA node the host buffer lives on?  (pinned buffers allocated and touched by a
thread bound to each node in turn)"""
import ctypes as C
import glob
import os
import time
import torch
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
B = 64 << 20
torch.cuda.init()
print("gpu pci:", torch.cuda.get_device_properties(0))
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(p, open(p).read().strip())
nodes = {}
for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    nodes[p.split("/")[-2]] = open(p).read().strip()
print(nodes)
dp = C.c_void_p(); assert hip.hipMalloc(C.byref(dp), B) == 0
s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1)


def cpus(spec):
    out = set()
    for part in spec.split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


all_cpus = os.sched_getaffinity(0)
for name, spec in nodes.items():
    want = cpus(spec) & all_cpus
    if not want:
        print(name, "no cpus allowed"); continue
    os.sched_setaffinity(0, want)
    hp = C.c_void_p(); assert hip.hipHostMalloc(C.byref(hp), B, 0) == 0
    C.memset(hp, 1, B)
    import numpy as np
    pg = np.empty(B, dtype=np.uint8); pg.fill(1)
    res = []
    for what, host in (("pinned", hp.value), ("pageable", pg.ctypes.data)):
        tu = td = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); hip.hipMemcpyAsync(dp, host, B, 1, s); hip.hipStreamSynchronize(s); tu = min(tu, time.perf_counter() - t0)
            t0 = time.perf_counter(); hip.hipMemcpyAsync(host, dp, B, 2, s); hip.hipStreamSynchronize(s); td = min(td, time.perf_counter() - t0)
        res.append(f"{what}: up {B / tu / 1e9:.1f} GB/s, down {B / td / 1e9:.1f} GB/s")
    print(name, "|", " | ".join(res), flush=True)
os.sched_setaffinity(0, all_cpus)
