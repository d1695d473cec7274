"""Fuzz the device-resident entries (zstdlite_amd.device.DeviceCodec: zl_compress_device_* / zl_decompress_device_*): random
sizes (the edges of chunks over-represented), both layouts, checksum, levels; the context is REUSED from call to call with
sizes going up and down (grow-only scratch, stale state).  The output must be what the host entry writes for the same bytes and
options, the reference must decode it, and the device decoder -- with and without the chunk hint -- must return the input.
usage: fuzz_device.py [rounds=200] [seed=1]"""
import sys
import time
import warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
import zstdlite_amd as z
from zstdlite_amd.device import DeviceCodec
from zl_testlib import GENERATORS, Ref
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ref = Ref()
warnings.simplefilter("ignore")
pool = {k: GENERATORS[k](8 << 20, seed=51) for k in ("text", "iid10", "int20", "rand")}
edges = [1, 2, 8, 9, 255, 131071, 131072, 131073, 262144, 262145, 1048576, 1048577, 4 << 20]
codecs = {}
bad = 0
t0 = time.time()
for rd in range(rounds):
    n = int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(1, 24 << 20))
    layout = str(rng.choice(["frames", "one_frame"]))
    cs = bool(rng.random() < 0.3)
    level = int(rng.choice([1, 3, 3, 5, 6, 9, -5, 16, 19]))
    key = (layout, cs, level)
    try:
        if key not in codecs:
            codecs[key] = DeviceCodec(level=level, include_checksum=cs, frame_layout=layout)
        codec = codecs[key]
        parts, have = [], 0
        while have < n:
            k = int(min(n - have, int(rng.integers(1, 3 << 20))))
            src = pool[str(rng.choice(list(pool)))]
            o = int(rng.integers(0, len(src) - k))
            parts.append(src[o:o + k]); have += k
        x = b"".join(parts)
        d_src = torch.from_numpy(np.frombuffer(x, dtype=np.uint8).copy()).cuda()
        dst, total, fs = codec.compress(d_src)
        comp = bytes(dst[:total].cpu().numpy())
        assert int(np.sum(fs)) == total, "frame sizes do not add up"
        host = z.zstd_compress(x, level=level, include_checksum=cs, frame_layout=layout)
        assert comp == host, "device entry and host entry write different bytes"
        assert ref.decompress(comp, n) == x, "reference decoder disagrees"
        back = torch.zeros(n, dtype=torch.uint8, device="cuda")
        if layout == "frames":
            tbl = codec.frame_table(fs, codec.chunk_content_sizes(n))
            codec.decompress(dst, tbl, back)
            assert bytes(back.cpu().numpy()) == x, "device decode differs"
        else:
            tbl = codec.frame_table([total], [n])
            for hint in (fs, None):
                back.zero_()
                codec.decompress(dst, tbl, back, chunk_sizes=hint)
                assert bytes(back.cpu().numpy()) == x, ("device decode differs", "hint" if hint is not None else "no hint")
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL round", rd, "n", n, key, repr(e)[:300], flush=True)
print(f"{rounds} rounds, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
