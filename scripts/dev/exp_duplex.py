"""PCIe: upload and download of pageable (and pinned) host buffers alone and at the same time, from two host threads."""
import ctypes as C, threading, time, sys
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
dev = torch.device("cuda", 0)
n_up, n_down = 660 << 20, 1 << 30
d_up = torch.empty(n_up, dtype=torch.uint8, device=dev); d_down = torch.empty(n_down, dtype=torch.uint8, device=dev)
h_up = np.random.randint(0, 255, n_up, dtype=np.uint8); h_down = np.zeros(n_down, dtype=np.uint8)
def up(): hip.hipMemcpy(d_up.data_ptr(), h_up.ctypes.data, n_up, 1)
def down(): hip.hipMemcpy(h_down.ctypes.data, d_down.data_ptr(), n_down, 2)
def timed(fns, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for label in ("pageable", "pinned"):
    if label == "pinned":
        print("register", hip.hipHostRegister(h_up.ctypes.data, n_up, 0), hip.hipHostRegister(h_down.ctypes.data, n_down, 0))
    a, b, c = timed([up]), timed([down]), timed([up, down])
    print(f"{label}: up {n_up/a/1e6:.1f} GB/s ({a:.1f} ms)  down {n_down/b/1e6:.1f} GB/s ({b:.1f} ms)  both at once {c:.1f} ms")
    # the same bytes in 8 pieces per direction, two threads
    def up8():
        for i in range(8): hip.hipMemcpy(d_up.data_ptr() + i * (n_up // 8), h_up.ctypes.data + i * (n_up // 8), n_up // 8, 1)
    def down8():
        for i in range(8): hip.hipMemcpy(h_down.ctypes.data + i * (n_down // 8), d_down.data_ptr() + i * (n_down // 8), n_down // 8, 2)
    print(f"   in 8 pieces each: up {timed([up8]):.1f} ms down {timed([down8]):.1f} ms both {timed([up8, down8]):.1f} ms")
# the same with hipMemcpyAsync on two non-blocking streams (the buffers are registered by now)
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
s1, s2 = C.c_void_p(), C.c_void_p()
hip.hipStreamCreateWithFlags(C.byref(s1), 1); hip.hipStreamCreateWithFlags(C.byref(s2), 1)
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
def upa(): hip.hipMemcpyAsync(d_up.data_ptr(), h_up.ctypes.data, n_up, 1, s1); hip.hipStreamSynchronize(s1)
def downa(): hip.hipMemcpyAsync(h_down.ctypes.data, d_down.data_ptr(), n_down, 2, s2); hip.hipStreamSynchronize(s2)
print(f"pinned, async on two streams: up {timed([upa]):.1f} ms down {timed([downa]):.1f} ms both {timed([upa, downa]):.1f} ms")
h2 = np.random.randint(0, 255, n_up, dtype=np.uint8); h3 = np.zeros(n_down, dtype=np.uint8)
def upb(): hip.hipMemcpyAsync(d_up.data_ptr(), h2.ctypes.data, n_up, 1, s1); hip.hipStreamSynchronize(s1)
def downb(): hip.hipMemcpyAsync(h3.ctypes.data, d_down.data_ptr(), n_down, 2, s2); hip.hipStreamSynchronize(s2)
print(f"pageable, async on two streams: up {timed([upb]):.1f} ms down {timed([downb]):.1f} ms both {timed([upb, downb]):.1f} ms")
# upload by a kernel that reads the registered host buffer (torch copy from a mapped pointer) while the engine downloads
dp = C.c_void_p()
hip.hipHostGetDevicePointer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
print("devptr", hip.hipHostGetDevicePointer(C.byref(dp), h_up.ctypes.data, 0))
hip.hipMemcpyDtoDAsync = hip.hipMemcpyAsync
def upk(): hip.hipMemcpyAsync(d_up.data_ptr(), dp, n_up, 3, s1); hip.hipStreamSynchronize(s1)   # device-to-device kind on the mapped pointer: a copy kernel
print(f"pinned, upload as a device-side copy of the mapped buffer: up {timed([upk]):.1f} ms  with the download beside it {timed([upk, downa]):.1f} ms")
