"""Dev: zl_decompressDCtx host to host on ONE frame of 1 GiB of the bench's text quarter (high ratio: a short upload)."""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed
L = _lib.lib()
src, kinds, quarter = gen_mixed(1 << 30, torch.device("cuda", 0))
i = [k for k, name in enumerate(kinds) if name == "text"][0]
q = src[i * quarter:(i + 1) * quarter].cpu().numpy()
host = np.concatenate([q, q, q, q])
n = host.size
cap = L.zl_compressBound(n)
comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
back = np.empty(n, dtype=np.uint8); back.fill(1)
cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
for it in range(4):
    r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
    t1 = time.perf_counter()
    r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
    print(f"--- iteration {it}: {r} bytes, decompress {(t2 - t1) * 1e3:.1f} ms = {n / (t2 - t1) / 1e9:.1f} GB/s", file=sys.stderr, flush=True)
assert r2 == n and np.array_equal(back, host)
