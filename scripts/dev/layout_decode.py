"""Dev: decode of the same input in the two layouts, device-resident, for a kernel trace: python layout_decode.py <kind|mix> <MiB> <frames|one_frame>"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
kind, mib, layout = sys.argv[1], int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda", 0)
n = mib << 20
src = gen_mixed(n, dev)[0] if kind == "mix" else gen_device(kind, n, dev)
codec = DeviceCodec(level=3, profile=True, frame_layout=layout)
comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
back = torch.empty(n, dtype=torch.uint8, device=dev)
_, total, fs = codec.compress(src, comp)
if layout == "one_frame":
    tbl = codec.frame_table([total], [n])
else:
    sizes = [int(x) for x in fs.cpu().tolist()]
    tbl = codec.frame_table(sizes, [min(131072, n - i * 131072) for i in range(len(sizes))])
for it in range(4):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    got, _ = codec.decompress(comp, tbl, back, chunk_sizes=fs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{kind} {mib} MiB {layout}: decompress {(t2 - t1) * 1e3:.2f} ms ({n / (t2 - t1) / 1e9:.1f} GB/s)", flush=True)
assert got == n and torch.equal(back, src)
