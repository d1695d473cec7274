"""Dev: levels 13 - 15 with and without the second search pass (build variant -DZL_BT_K2=...): size against the reference at the level, host call time."""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref
ref = Ref()
inputs = {"corpus": LONG_RANGE["corpus"](None), "text16M": GENERATORS["text"](16 << 20), "records8M": LONG_RANGE["records"](8 << 20)}
for name, data in inputs.items():
    for lvl in [int(x) for x in sys.argv[1].split(",")]:
        cctx = z.zstd_cctx(level=lvl)
        comp = z.zstd_compress(data, cctx=cctx)
        t0 = time.time(); comp = z.zstd_compress(data, cctx=cctx); dt = time.time() - t0
        assert ref.decompress(comp, len(data)) == data
        theirs = len(ref.compress(data, level=lvl))
        print(f"{name:9s} level {lvl:2d} ours {len(comp):9d} ref {theirs:9d} size ratio {len(comp)/theirs:.3f} {dt*1e3:7.1f} ms", flush=True)
