"""Dev: the entropy kernel's in-kernel phase stamps (ZL_c_profile 2) for ONE small input: where a chunk's 0.4 - 0.65 ms go."""
import sys, ctypes as C, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS
L = _lib.lib()
L.zl_cctx_phase_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for kind in ("text", "int20"):
    for kib in (64, 1024):
        n = kib << 10
        data = GENERATORS[kind](n, seed=5)
        cctx = z.zstd_cctx(level=3)
        L.zl_CCtx_setParameter(cctx._ptr, 9001, 2)
        cap = L.zl_compressBound(n); out = (C.c_char * cap)()
        for _ in range(3): r = L.zl_compress2(cctx._ptr, out, cap, data, n)
        ct = (C.c_ulonglong * 16)()
        assert L.zl_cctx_phase_stats(cctx._ptr, ct) == 0
        nc = max(ct[14], 1)
        print(f"{kind} {kib} KiB: chunks {ct[14]}  kcycles/chunk: statistics {ct[8]/nc/1e3:.0f}  tree|tables {ct[9]/nc/1e3:.0f}  codes+bits {ct[10]/nc/1e3:.0f}  chains|tree description {ct[13]/nc/1e3:.0f}  blk literals {ct[11]/nc/1e3:.0f}  blk sequences {ct[12]/nc/1e3:.0f}")
