"""bench.py's end_to_end leg alone (the host entries on the 1 GiB mix, both layouts), for runs under a development build with
ZL_DEBUG_PIPE / ZL_DEBUG_SLAB.  usage: e2e_only.py"""
import json
import sys
sys.path.insert(0, ".")
import torch
import bench
from zstdlite_amd.device import gen_mixed
dev = torch.device("cuda", 0)
host = gen_mixed(1 << 30, dev)[0].cpu().numpy()
out = bench.end_to_end_leg(host, dev, lambda *a: None)
print(json.dumps({k: out[k] for k in ("one_frame", "independent_frames")}))
