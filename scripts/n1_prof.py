"""N1 under the profiler: the reference's single frame of the bench mix decoded device-resident by the
large-frame pipeline.  usage: n1_prof.py [MiB]   (rocprofv3 --kernel-trace --stats -- python3 scripts/n1_prof.py 512)"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
from zstdlite_amd import _lib
from zl_testlib import Ref
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
src, kinds, quarter = gen_mixed(mb << 20, dev)
host = src.cpu().numpy()
t0 = time.time(); frame = Ref().compress(host.tobytes(), level=3); print(f"reference frame: {len(frame)} bytes in {time.time() - t0:.1f}s", flush=True)
codec = DeviceCodec(level=3, profile=True)
d_src = torch.from_numpy(np.frombuffer(frame, dtype=np.uint8).copy()).to(dev)
d_dst = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
tbl = codec.frame_table([len(frame)], [mb << 20])
for it in range(3):
    d_dst.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); total, _ = codec.decompress(d_src, tbl, d_dst); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"decode {it}: {dt * 1e3:.2f} ms, {(mb << 20) / dt / 1e9:.2f} GB/s", flush=True)
assert torch.equal(d_dst, src) and _lib.lib().zl_dctx_large_frames(codec.dctx) == 3
