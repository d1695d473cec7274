"""N1: decode of one large reference-made frame (many blocks, one window) -- where the general kernel stands."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import Ref, GENERATORS
ref = Ref()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for gen in ("text", "iid10", "int20", "rand"):
    data = GENERATORS[gen](mb << 20, seed=5)
    t0 = time.time(); comp = ref.compress(data, level=3); t_c = time.time() - t0
    t0 = time.time(); back = ref.decompress(comp, len(data)); t_cpu = time.time() - t0
    assert back == data
    dctx = z.zstd_dctx()
    z.zstd_decompress(comp, dctx=dctx)
    t0 = time.time(); out = z.zstd_decompress(comp, dctx=dctx); t_gpu = time.time() - t0
    assert out == data
    print(f"{gen}: {mb} MiB, ratio {len(data) / len(comp):.2f}; cpu decode {len(data) / t_cpu / 1e9:.2f} GB/s, ours (host to host) {len(data) / t_gpu / 1e9:.3f} GB/s", flush=True)
