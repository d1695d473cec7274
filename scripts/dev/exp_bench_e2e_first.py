"""The end_to_end leg of bench.py as the first thing a process does (is the link used both ways then?), and again after the
other legs' typical work."""
import sys, json, importlib.util
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); sys.argv = ["bench.py"]; spec.loader.exec_module(b)
from zstdlite_amd.device import DeviceCodec, gen_mixed
dev = torch.device("cuda", 0)
n = 1 << 30
src, _, _ = gen_mixed(n, dev)
host = src.cpu().numpy()
log = lambda *a: None
def show(tag, e): print(tag, {k: (round(v["compress_GBps"], 1), round(v["decompress_GBps"], 1)) for k, v in e.items() if isinstance(v, dict) and "compress_GBps" in v}, "both_at_once_ms", round(e["pcie_GBps_measured"].get("both_at_once_ms", -1), 1))
show("first:", b.end_to_end_leg(host, dev, log))
# what the headline loop does: device-resident compress / decompress on a DeviceCodec
c = DeviceCodec()
comp = torch.empty(c.bound(n), dtype=torch.uint8, device=dev)
for _ in range(3):
    _, total, fs = c.compress(src, comp)
    back = torch.empty_like(src)
    c.decompress(comp, c.frame_table(fs, c.chunk_content_sizes(n)), back)
show("after device-resident round trips:", b.end_to_end_leg(host, dev, log))
