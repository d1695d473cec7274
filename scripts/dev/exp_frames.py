"""Decode N reference-made frames of S bytes each (text / iid10 alternating); prints GB/s."""
import sys, time, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device
from zl_testlib import Ref
dev = torch.device("cuda", 0)
S = int(sys.argv[1]); N = int(sys.argv[2])
ref = Ref(); L = ref.L
srcs = {k: gen_device(k, (N // 2 + 1) * S, dev, seed=5).cpu().numpy() for k in ("text", "iid10")}
cctx = L.ZSTD_createCCtx(); L.ZSTD_CCtx_setParameter(cctx, 100, 3)
cap = L.ZSTD_compressBound(S); buf = C.create_string_buffer(cap)
frames, plain = [], []
for i in range(N):
    a = srcs[("text", "iid10")[i % 2]][(i // 2) * S:(i // 2 + 1) * S]
    r = L.ZSTD_compress2(cctx, buf, cap, a.ctypes.data, S); frames.append(buf.raw[:r]); plain.append(a)
blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
want = torch.from_numpy(np.concatenate(plain)).to(dev)
d_src = torch.from_numpy(blob.copy()).to(dev); d_dst = torch.empty(N * S, dtype=torch.uint8, device=dev)
codec = DeviceCodec(profile=True)
tbl = codec.frame_table([len(f) for f in frames], [S] * N)
best = 1e9
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    total, _ = codec.decompress(d_src, tbl, d_dst)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
assert torch.equal(d_dst, want)
print(f"{N} frames x {S >> 10} KiB: {N * S / best / 1e9:.2f} GB/s", {k: round(v, 2) for k, v in codec.kernel_ms().items() if k.startswith("d")})
