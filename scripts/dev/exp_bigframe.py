"""N1: decode of one large reference-made frame (many blocks, one window): the large-frame pipeline
(zl_decode3.hip) against the reference library on the host.  ZL_DEBUG_BIG=1 prints the stage times,
ZL_NO_BIG=1 sends the frame through the general kernel instead.
usage: exp_bigframe.py [MiB] [kinds,comma-separated] [level]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
from zl_testlib import Ref, GENERATORS
ref = Ref()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["text", "iid10", "int20", "rand"]
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for gen in kinds:
    data = GENERATORS[gen](mb << 20, seed=5)
    t0 = time.time(); comp = ref.compress(data, level=level); t_c = time.time() - t0
    t0 = time.time(); back = ref.decompress(comp, len(data)); t_cpu = time.time() - t0
    assert back == data
    dctx = z.zstd_dctx()
    assert z.zstd_decompress(comp, dctx=dctx) == data
    # the C-ABI call itself, as the reference's C glue would make it (src/raw-file.c:189): host buffers in, host buffer out
    import ctypes as C, numpy as np
    from zstdlite_amd import _lib
    src_np = np.frombuffer(comp, dtype=np.uint8); out_np = np.zeros(len(data), dtype=np.uint8)
    t_gpu = 1e9
    for _ in range(3):
        t0 = time.time(); r = _lib.lib().zl_decompressDCtx(dctx._ptr, C.c_void_p(out_np.ctypes.data), len(data), C.c_void_p(src_np.ctypes.data), len(comp)); t_gpu = min(t_gpu, time.time() - t0)
    assert r == len(data) and out_np.tobytes() == data
    print(f"{gen}: {mb} MiB, level {level}, ratio {len(data) / len(comp):.2f}; cpu decode {len(data) / t_cpu / 1e9:.2f} GB/s, ours (host to host) {len(data) / t_gpu / 1e9:.3f} GB/s", flush=True)
