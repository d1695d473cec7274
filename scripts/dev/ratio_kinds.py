"""Compressed sizes of the bench kinds, the source-text corpus and a pickled-objects blob, both layouts (A/B of parse rules)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device
from zl_testlib import gen_corpus, gen_records
dev = torch.device("cuda", 0)
inputs = {k: gen_device(k, 64 << 20, dev) for k in ("iid10", "text", "int20")}
inputs["corpus"] = torch.from_numpy(np.frombuffer(gen_corpus(), dtype=np.uint8).copy()).to(dev)
inputs["records"] = torch.from_numpy(np.frombuffer(gen_records(32 << 20), dtype=np.uint8).copy()).to(dev)
for layout in ("frames", "one_frame"):
    c = DeviceCodec(frame_layout=layout)
    row = []
    for k, src in inputs.items():
        comp, total, fs = c.compress(src)
        row.append(f"{k} {src.numel() / total:.4f}")
    print(layout, " | ".join(row))
    c.close()
