"""Dev: ONE frame this library wrote, decoded a few times through the C entry (for a kernel trace)."""
import sys, time, ctypes as C, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS
L = _lib.lib()
kind, n = sys.argv[1], int(float(sys.argv[2]) * 1024)
data = GENERATORS[kind](n, seed=5)
frame = z.zstd_compress(data)
dctx = z.zstd_dctx()
out = (C.c_char * n)()
for rep in range(5):
    t0 = time.perf_counter(); r = L.zl_decompressDCtx(dctx._ptr, out, n, frame, len(frame)); dt = time.perf_counter() - t0
    print(f"{dt * 1e3:.3f} ms", flush=True)
assert r == n and bytes(out) == data
