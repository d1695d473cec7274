"""Dev: ONE small input compressed a few times through the C entry (for a kernel trace)."""
import sys, time, ctypes as C, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS
L = _lib.lib()
kind, n, lvl = sys.argv[1], int(float(sys.argv[2]) * 1024), int(sys.argv[3]) if len(sys.argv) > 3 else 3
data = GENERATORS[kind](n, seed=5)
cctx = z.zstd_cctx(level=lvl)
cap = L.zl_compressBound(n); out = (C.c_char * cap)()
for rep in range(5):
    t0 = time.perf_counter(); r = L.zl_compress2(cctx._ptr, out, cap, data, n); dt = time.perf_counter() - t0
    print(f"{dt * 1e3:.3f} ms -> {r} bytes", flush=True)
