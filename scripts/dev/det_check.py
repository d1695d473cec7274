"""Dev: is the encoder deterministic and the round trip exact at 1 GiB?  usage: det_check.py [profile=0] [kinds]"""
import sys, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import CHUNK, DeviceCodec, gen_device
dev = torch.device("cuda", 0)
n = 1 << 30
prof = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["iid10", "iid10", "text", "rand", "iid10", "text"]
codec = DeviceCodec(profile=bool(prof))
seen = {}
for kind in kinds:
    src = gen_device(kind, n, dev)
    comp, total, fs = codec.compress(src)
    back = torch.empty_like(src)
    tbl = codec.frame_table(fs, codec.chunk_content_sizes(n))
    got, res = codec.decompress(comp, tbl, back)
    eq = torch.equal(back, src)
    bad_chunks = []
    if not eq:
        bad = torch.nonzero(back != src).flatten()
        bad_chunks = sorted(set((bad // CHUNK).tolist()))
    tag = (kind, total)
    print(kind, "total", total, "equal", eq, "previous totals", seen.get(kind), "bad chunks", len(bad_chunks), bad_chunks[:12], flush=True)
    seen.setdefault(kind, []).append(total)
