"""Dev: where does a reference frame written at a high level go, and how fast?  (level-19 frames on the general kernel)
Device-resident decode of ONE reference-made frame, as bench.py's N1 leg.  usage: python scripts/dev/l19_probe.py [MiB=64] [levels=3,12,19]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import DeviceCodec
from zl_testlib import GENERATORS, LONG_RANGE, Ref
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
levels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,12,19").split(",")]
ref = Ref()
q = (mib << 20) // 4
corpus = LONG_RANGE["corpus"](None)
data = b"".join(GENERATORS[k](q) for k in ("iid10", "text", "int20")) + (corpus * (q // len(corpus) + 1))[:q]
dev = torch.device("cuda", 0)
n = len(data)
plain = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
for level in levels:
    t0 = time.time(); frame = np.frombuffer(ref.compress(data, level=level), dtype=np.uint8).copy(); t1 = time.time()
    codec = DeviceCodec(profile=True)
    d_src = torch.from_numpy(frame).to(dev)
    d_dst = torch.empty(n, dtype=torch.uint8, device=dev)
    tbl = codec.frame_table([int(frame.size)], [n])
    best = 1e9
    for it in range(4):
        d_dst.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        total, res = codec.decompress(d_src, tbl, d_dst)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    ok = total == n and torch.equal(d_dst, plain)
    km = codec.kernel_ms()
    print(f"level {level:2d}: {n} -> {frame.size} bytes ({t1 - t0:.1f}s on the host); decode {n / best / 1e9:7.2f} GB/s ({best * 1e3:.2f} ms) exact {ok}; large frames {_lib.lib().zl_dctx_large_frames(codec.dctx)}; "
          + " ".join(f"{k} {v:.2f}" for k, v in km.items() if k.startswith("d")), flush=True)
    codec.close()
