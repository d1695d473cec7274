"""Ratio and compress time per level, both layouts (device-resident), next to the reference at the same level."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device
from zl_testlib import gen_corpus, gen_records, Ref, have_ref
dev = torch.device("cuda", 0)
inputs = {"text": gen_device("text", 64 << 20, dev), "iid10": gen_device("iid10", 64 << 20, dev),
          "corpus": torch.from_numpy(np.frombuffer(gen_corpus(), dtype=np.uint8).copy()).to(dev),
          "records": torch.from_numpy(np.frombuffer(gen_records(32 << 20), dtype=np.uint8).copy()).to(dev)}
ref = Ref() if have_ref() else None
for level in (1, 3, 5):
    for layout in ("frames", "one_frame"):
        c = DeviceCodec(level=level, frame_layout=layout)
        row = []
        for k, src in inputs.items():
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); comp, total, fs = c.compress(src); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            row.append(f"{k} {src.numel() / total:.3f} ({src.numel() / best / 1e9:.0f} GB/s)")
        print(f"level {level} {layout:9s}", " | ".join(row))
        c.close()
    if ref is not None:
        print(f"level {level} reference", " | ".join(f"{k} {src.numel() / len(ref.compress(src.cpu().numpy().tobytes(), level=level)):.3f}" for k, src in inputs.items() if k != "iid10"))
