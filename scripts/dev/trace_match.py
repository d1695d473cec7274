"""Dev: timeline of one match workgroup (library built with -DZL_TRACE=<chunk>): per wave and iteration the cycle
stamps the lean stride-1 body takes.  usage: ZL_HIP_LIB=<trace variant> python scripts/dev/trace_match.py [kind] [MiB]"""
import ctypes as C, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
dev = torch.device("cuda", 0)
src = gen_mixed(n, dev)[0] if kind == "mix" else gen_device(kind, n, dev)
import os
codec = DeviceCodec(profile=2, frame_layout=os.environ.get("ZL_LAYOUT", "frames"))
for _ in range(2):
    codec.compress(src)
POINTS = 16
words = (1 << 20) // 8
buf = (C.c_ulonglong * words)()
codec.L.zl_cctx_trace.restype = C.c_size_t
got = codec.L.zl_cctx_trace(C.c_void_p(codec.cctx), buf, words)
a = np.frombuffer(buf, dtype=np.uint64)[: (words // (4 * POINTS)) * 4 * POINTS].reshape(-1, 4, POINTS).astype(np.int64)
its = int((a[:, 0, 0] != 0).sum())
a = a[8:its - 2]  # steady state
names = ["top", "lookup", "gather+cmp", "extend", "barA", "closure", "barB", "chain", "marks", "scan+barC", "emit", "insert", "barG"]
print(f"{kind}: {its} stamped iterations; cycles between stamps, mean over iterations (per wave 0..3), then time of an iteration")
for k in range(1, 13):
    d = a[:, :, k] - a[:, :, k - 1]
    print(f"  {names[k - 1]:>10s} -> {names[k]:<10s} " + " ".join(f"{d[:, w].mean():7.0f}" for w in range(4)) + f"   max-wave mean {d.max(axis=1).mean():7.0f}")
whole = a[1:, 0, 0] - a[:-1, 0, 0]
print(f"  iteration period (wave 0 top to top): mean {whole.mean():.0f}  median {np.median(whole):.0f}")
# when does each wave ARRIVE at barrier A relative to the first arrival, and the spread
arr = a[:, :, 3]
print(f"  arrival spread at barrier A (last - first wave): mean {(arr.max(axis=1) - arr.min(axis=1)).mean():.0f}")
arr = a[:, :, 5]
print(f"  arrival spread at barrier B: mean {(arr.max(axis=1) - arr.min(axis=1)).mean():.0f}")
