import sys, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import zstdlite_amd as z
from zl_testlib import *
o = Oracle(); r = Ref() if have_ref() else None
for gen in ["text", "iid10", "int20", "rand"]:
    for n in [0, 1, 8, 65, 300, 1000, 4096, 65536, 131072, 131073, 1 << 20]:
        data = GENERATORS[gen](n, seed=n + 17)
        try:
            comp = z.zstd_compress(data)
        except Exception as e:
            print(gen, n, "COMPRESS FAIL", e); continue
        code, out = o.decompress_code(comp, n)
        ok = (not is_error(code)) and out == data
        msg = "" if ok else (o.error_name(code) if is_error(code) else "MISMATCH at %d" % next((i for i in range(min(len(out), n)) if out[i] != data[i]), -1))
        extra = ""
        if r is not None and n >= 4096:
            extra = " ref=%d" % len(r.compress(data, 3))
        print(gen, n, len(comp), "OK" if ok else "BAD " + msg, extra)
