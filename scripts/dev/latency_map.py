"""Dev: wall time of the host entries over sizes and options, next to the reference on one core -- to find cliffs (a route that is
right for a batch of thousands and wrong for ONE object).  Flags anything slower than 10 ms that is also 8x slower than the reference."""
import sys, time, ctypes as C, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref
ref = Ref(); L = _lib.lib()
t_ = GENERATORS["text"](1_300_000, seed=77)
D = ref.train_dict([t_[i * 3000:(i + 1) * 3000] for i in range(400)], 20000)
def best_of(f, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t0)
    return b, r
sizes = [3000, 40_000, 131072, 131073, 300_000, 1 << 20, (1 << 20) + 1, 3 << 20, 12 << 20]
kinds = sys.argv[1:] or ["text", "int20"]
for kind in kinds:
    for n in sizes:
        data = GENERATORS[kind](n, seed=3)
        cases = []
        for name, ckw, dkw in (("plain", {}, {}), ("checksum", {"include_checksum": True}, {}), ("dict", {"dict": D}, {"dict": D}), ("frames", {"frame_layout": "frames"}, {}), ("level9", {"level": 9}, {}), ("level1", {"level": 1}, {})):
            cctx = z.zstd_cctx(**ckw); dctx = z.zstd_dctx(**dkw)
            tc, comp = best_of(lambda: z.zstd_compress(data, cctx=cctx))
            td, back = best_of(lambda: z.zstd_decompress(comp, dctx=dctx))
            assert back == data
            rkw = {"level": ckw.get("level", 3)}
            if "dict" in ckw: rkw["dict"] = D
            if "include_checksum" in ckw: rkw["checksum"] = True
            trc, rcomp = best_of(lambda: ref.compress(data, **rkw), 1)
            trd, _ = best_of(lambda: ref.decompress(rcomp, n, **({"dict": D} if "dict" in ckw else {})), 1)
            tdr, back2 = best_of(lambda: z.zstd_decompress(rcomp, dctx=dctx))
            assert back2 == data
            flag = "".join(" <-- SLOW " + w for w, t, tr in (("compress", tc, trc), ("decompress own", td, trd), ("decompress reference-made", tdr, trd)) if t > 0.010 and t > 8 * tr)
            print(f"{kind:6s} {n:9d} {name:9s} compress {tc*1e3:7.2f} ms (ref {trc*1e3:7.2f})  decompress own {td*1e3:7.2f} / reference-made {tdr*1e3:7.2f} ms (ref {trd*1e3:7.2f}){flag}", flush=True)
