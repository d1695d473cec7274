import struct, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref, is_error
ref = Ref(); L = _lib.lib()
text = GENERATORS["text"](3 << 20, seed=41)
ints = GENERATORS["int20"](1 << 20, seed=42)
big_ref, big_own = ref.compress(text, level=3, checksum=True), z.zstd_compress(text)
small = ref.compress(ints[:5000], level=3)
for slab in (1, 0):
    dctx = z.zstd_dctx()
    if slab: assert L.zl_DCtx_setParameter(dctx._ptr, 9010, 1) == 0
    for name, buf in (("ref+3", big_ref + b"\x00\x01\x02"), ("ref-1", big_ref[:-1]), ("own/2", big_own[:len(big_own) // 2]), ("ref+small-3", big_ref + small[:-3]), ("own+magic", big_own + b"\x28\xb5\x2f\xfd")):
        cap = 2 * len(text) + 5000
        back = np.zeros(cap, dtype=np.uint8)
        r = L.zl_decompressDCtx(dctx._ptr, back.ctypes.data, cap, buf, len(buf))
        rc, _ = ref.decompress_code(buf, cap)
        try:
            m = len(z.zstd_decompress(buf, dctx=dctx))
        except z.ZstdError as e:
            m = "raises " + str(e)[:40]
        print(slab, name, "ours", _lib.error_name(r) if is_error(r) else r, "| ref", "error" if is_error(rc) else rc, "| mirror", m)
