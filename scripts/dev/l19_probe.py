"""Dev: where does a reference frame written at a high level go?  (level-19 frames on the general kernel)
usage: python scripts/dev/l19_probe.py [MiB=32] [level=19]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, LONG_RANGE, Ref
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
level = int(sys.argv[2]) if len(sys.argv) > 2 else 19
ref = Ref()
q = (mib << 20) // 4
data = b"".join(GENERATORS[k](q) for k in ("iid10", "text", "int20")) + LONG_RANGE["corpus"](None)[:q]
t0 = time.time(); frame = ref.compress(data, level=level); t1 = time.time()
print(f"reference level {level}: {len(data)} -> {len(frame)} bytes in {t1 - t0:.1f}s")
d = z.zstd_dctx()
L = _lib.lib()
for it in range(3):
    t0 = time.time(); back = z.zstd_decompress(frame, dctx=d); dt = time.time() - t0
assert back == data
import ctypes as C
ms = (C.c_double * 5)()
L.zl_dctx_kernel_ms(d._ptr, ms)
print(f"decompress {len(data) / dt / 1e9:.3f} GB/s (host entry); large frames {L.zl_dctx_large_frames(d._ptr)}; kernel ms decode {ms[0]:.2f} index {ms[1]:.2f} entropy {ms[2]:.2f} exec {ms[3]:.2f} general {ms[4]:.2f}")
