"""Dev: frames THIS library wrote (host entry: one frame), host memory to host memory through the C entry: wall time by size."""
import sys, time, ctypes as C, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS
L = _lib.lib()
dctx = z.zstd_dctx()
for kind in sys.argv[1:] or ["text"]:
    for mib in (0.0625, 0.125, 0.25, 0.5, 1, 2, 4, 16, 64):
        n = int(mib * (1 << 20))
        data = GENERATORS[kind](n, seed=5)
        for lvl in (3,):
            frame = z.zstd_compress(data, level=lvl)
            out = (C.c_char * n)()
            best = 1e9
            for rep in range(4):
                t0 = time.perf_counter(); r = L.zl_decompressDCtx(dctx._ptr, out, n, frame, len(frame)); best = min(best, time.perf_counter() - t0)
            assert r == n and bytes(out) == data
            print(f"{kind} {mib:7.3f} MiB level {lvl}: decompress {best * 1e3:7.2f} ms = {n / best / 1e9:6.2f} GB/s   large frames so far {int(L.zl_dctx_large_frames(dctx._ptr))}", flush=True)
