"""Dev: one reference-made frame from host memory to host memory through the C entry alone (output buffer made in front): wall time
by size, next to the reference's own decoder on one core."""
import sys, time, ctypes as C, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref
ref = Ref(); L = _lib.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
dctx = z.zstd_dctx()
cctx = z.zstd_cctx()
for mib in (0.25, 1, 4, 16, 32, 64, 256):
    n = int(mib * (1 << 20))
    data = GENERATORS[kind](n, seed=5)
    frame = ref.compress(data, level=3)
    out = (C.c_char * n)()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); r = L.zl_decompressDCtx(dctx._ptr, out, n, frame, len(frame)); best = min(best, time.perf_counter() - t0)
    assert r == n and bytes(out) == data
    t0 = time.perf_counter(); back = ref.decompress(frame, n); tr = time.perf_counter() - t0
    cap = L.zl_compressBound(n)
    comp = (C.c_char * cap)()
    bc = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); rc = L.zl_compress2(cctx._ptr, comp, cap, data, n); bc = min(bc, time.perf_counter() - t0)
    t0 = time.perf_counter(); rr = ref.compress(data, level=3); tc = time.perf_counter() - t0
    print(f"{kind} {mib:6.2f} MiB: decompress {best * 1e3:7.2f} ms = {n / best / 1e9:6.2f} GB/s (reference, one core: {tr * 1e3:7.2f} ms)   compress {bc * 1e3:7.2f} ms = {n / bc / 1e9:6.2f} GB/s (reference: {tc * 1e3:7.2f} ms)", flush=True)
