import sys; sys.path.insert(0,".")
import torch; torch.cuda.init(); torch.zeros(1, device="cuda")
from zstdlite_amd import _lib
L=_lib.lib()
import ctypes as C
for name, fn, w in (("exec", L.zl2_debug_occupancy, 0), ("lit", L.zl2_debug_occupancy, 1), ("seq", L.zl2_debug_occupancy, 2), ("match", L.zl_debug_occupancy, 0), ("entropy-enc", L.zl_debug_occupancy, 1)):
    fn.restype = C.c_int; print(name, fn(w))
