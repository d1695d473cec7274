"""Do two builds of the library write the same bytes?  (development: scheduling-only changes to the encoder must not change output)
usage: ZL_HIP_LIB=<variant .so> python scripts/dev/same_bytes.py write|check <file>"""
import hashlib, json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import zstdlite_amd as z
from zstdlite_amd.device import DeviceCodec, gen_mixed
from zl_testlib import GENERATORS, LONG_RANGE
from ratio_inputs import INPUTS
out = {}
dev = torch.device("cuda", 0)
src = gen_mixed(256 << 20, dev)[0]
for layout in ("frames", "one_frame"):
    c = DeviceCodec(frame_layout=layout)
    dst, total, fs = c.compress(src)
    out[f"mix256_{layout}"] = hashlib.sha256(dst[:total].cpu().numpy().tobytes()).hexdigest()
    c.close()
for name, gen in INPUTS.items():
    for n in (1024, 65536, 1 << 20):
        out[f"{name}_{n}"] = hashlib.sha256(z.zstd_compress(gen(n))).hexdigest()
out["corpus"] = hashlib.sha256(z.zstd_compress(LONG_RANGE["corpus"](None))).hexdigest()
out["corpus_l5_frames"] = hashlib.sha256(z.zstd_compress(LONG_RANGE["corpus"](None), level=5, frame_layout="frames")).hexdigest()
out["text_l1"] = hashlib.sha256(z.zstd_compress(GENERATORS["text"](3 << 20), level=1)).hexdigest()
out["special"] = hashlib.sha256(z.zstd_compress(bytes(300000) + b"ab" * 70000 + bytes(range(256)) * 500, include_checksum=True)).hexdigest()
if sys.argv[1] == "write":
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print("written", len(out))
else:
    want = json.load(open(sys.argv[2]))
    bad = [k for k in want if want[k] != out.get(k)]
    print("SAME BYTES" if not bad else f"DIFFERENT: {bad}")
