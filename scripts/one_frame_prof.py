"""The DEFAULT layout of the host entries (ONE frame, include/zl_hip.h) under the profiler: compress + decompress of the 1 GiB
bench mix, device-resident, a few times.  usage: rocprofv3 --kernel-trace --stats -- python3 scripts/one_frame_prof.py [MiB] [reps] [level]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
n = mib << 20
src = gen_mixed(n, dev)[0]
codec = DeviceCodec(level=level, profile=True, frame_layout="one_frame")
comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
back = torch.empty(n, dtype=torch.uint8, device=dev)
for it in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, total, fs = codec.compress(src, comp)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    got, _ = codec.decompress(comp, codec.frame_table([total], [n]), back, chunk_sizes=fs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"one frame {mib} MiB: ratio {n / total:.4f} compress {(t1 - t0) * 1e3:.2f} ms ({n / (t1 - t0) / 1e9:.1f} GB/s) decompress {(t2 - t1) * 1e3:.2f} ms ({n / (t2 - t1) / 1e9:.1f} GB/s) kernels {codec.kernel_ms()}", flush=True)
assert got == n and torch.equal(back, src)
