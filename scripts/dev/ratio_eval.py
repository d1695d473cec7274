"""Prints our compressed size next to the reference's level-3 size for the SURVEY 8(d) inputs."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import zstdlite_amd as z
from zl_testlib import *
r = Ref() if have_ref() else None
n = int(sys.argv[1]) if len(sys.argv) > 1 else (8 << 20)
for gen in ["text", "iid10", "int20", "rand"]:
    data = GENERATORS[gen](n)
    t0 = time.time(); ours = z.zstd_compress(data); t1 = time.time()
    ok = z.zstd_decompress(ours) == data
    theirs = len(r.compress(data, 3)) if r else 0
    print(f"{gen:6s} n={n} ours={len(ours)} ref={theirs} ours/ref={len(ours)/max(theirs,1):.4f} ratio={n/len(ours):.3f} roundtrip={ok} t={t1-t0:.3f}s")
