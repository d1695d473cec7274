"""What the content checksum costs the device-resident decode of ONE frame: XXH64 is one chain through the whole content
(zl_device.h zl_xxh64_low32), a GPU lane's work.  usage: checksum_cost.py [MiB=1024]"""
import sys
import time
sys.path.insert(0, ".")
import torch
from zstdlite_amd.device import DeviceCodec, gen_mixed
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
dev = torch.device("cuda", 0)
src = gen_mixed(n, dev)[0]
for validate in (True, False):
    codec = DeviceCodec(frame_layout="one_frame", include_checksum=True, validate_checksum=validate)
    comp, total, fs = codec.compress(src)
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        codec.decompress(comp, codec.frame_table([total], [n]), out, chunk_sizes=fs)
        torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    assert torch.equal(out, src)
    print(f"{mib} MiB one frame with checksum, validate={validate}: {best * 1e3:.1f} ms ({n / best / 1e9:.1f} GB/s)")
    codec.close()
