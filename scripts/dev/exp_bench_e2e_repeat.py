import sys, importlib.util
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); sys.argv = ["bench.py"]; spec.loader.exec_module(b)
from zstdlite_amd.device import DeviceCodec, gen_mixed
dev = torch.device("cuda", 0)
n = 1 << 30
src, _, _ = gen_mixed(n, dev)
host = src.cpu().numpy()
log = lambda *a: None
def show(tag, e): print(tag, {k: (round(v["compress_GBps"], 1), round(v["decompress_GBps"], 1)) for k, v in e.items() if isinstance(v, dict) and "compress_GBps" in v}, flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
for k in range(3):
    show(f"{mode} {k}:", b.end_to_end_leg(host, dev, log))
    if mode == "codec":   # a device-resident codec is created, used and CLOSED between the legs
        c = DeviceCodec(); comp = torch.empty(c.bound(n), dtype=torch.uint8, device=dev); c.compress(src, comp); c.close(); del comp
    if mode == "codec_alive":
        c = DeviceCodec(); comp = torch.empty(c.bound(n), dtype=torch.uint8, device=dev); c.compress(src, comp); del comp
