"""Generates tests/golden/frames.npz: zstd frames produced by the REFERENCE libzstd 1.5.6
(oracle/_ref/libzstd_ref.so, compiled from /root/reference/src/zstd/zstd.c by
`make -C oracle ref`) through the reference glue's own call sequence
(ZSTD_compressBound -> ZSTD_compress2, src/raw-file.c:52-74), plus the SHA-256 of each
plaintext.  Plaintexts are regenerated from (generator, n) by tests/zl_testlib.py, so only
compressed bytes + hashes are stored.  Run in the build container only:

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zl_testlib import GENERATORS, GOLDEN, Ref, sha256  # noqa: E402

CASES = []
for gen in ("text", "iid10", "int20", "rand"):
    for n in (0, 1, 7, 100, 1000, 4096, 65536, 131072, 200000):
        if gen == "rand" and n > 4096:
            continue
        for level, checksum in ((3, False), (1, False), (9, True), (19, False), (-5, False)):
            if n > 65536 and level not in (3, 19):
                continue
            CASES.append((gen, n, level, checksum))


def main():
    ref = Ref()
    assert ref.version() == "1.5.6"
    blobs, meta = {}, []
    for i, (gen, n, level, checksum) in enumerate(CASES):
        data = GENERATORS[gen](n)
        comp = ref.compress(data, level=level, checksum=checksum)
        key = f"f{i:03d}"
        blobs[key] = np.frombuffer(comp, dtype=np.uint8)
        meta.append({"key": key, "gen": gen, "n": n, "level": level, "checksum": checksum,
                     "csize": len(comp), "sha256": sha256(data)})
    # dictionary case (C4 shape): samples = small serialised-looking records
    samples = [(b"name=%d;rank=" % i) + bytes(str((i * 7919) % 50), "ascii") * 3 + GENERATORS["text"](120, seed=i)
               for i in range(2000)]
    d = ref.train_dict(samples, 4096)
    blobs["dict"] = np.frombuffer(d, dtype=np.uint8)
    for j in range(8):
        comp = ref.compress(samples[j], level=3, dict=d)
        key = f"d{j:03d}"
        blobs[key] = np.frombuffer(comp, dtype=np.uint8)
        blobs[key + "_plain"] = np.frombuffer(samples[j], dtype=np.uint8)
        meta.append({"key": key, "gen": "dict", "n": len(samples[j]), "level": 3, "checksum": False,
                     "csize": len(comp), "sha256": sha256(samples[j])})
    blobs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, "frames.npz"), **blobs)
    print(len(meta), "frames,", sum(m["csize"] for m in meta), "compressed bytes")


# Large frames: ONE reference-made frame of more than eight blocks each -- treeless literals and
# Repeat_Mode tables across blocks, matches across blocks, a checksum, runs (RLE blocks) and raw blocks.
LARGE = [  # (name, [(generator, n, seed)...] concatenated, level, checksum)
    ("text_l3", [("text", 1200000, 31)], 3, False),
    ("iid10_l3_ck", [("iid10", 1190000, 32)], 3, True),
    ("mixed_l19", [("text", 500000, 33), ("zeros", 300000, 0), ("rand", 150000, 34), ("int20", 300000, 35), ("text", 200000, 33)], 19, False),
]


def large_plain(parts):
    return b"".join(bytes(n) if g == "zeros" else GENERATORS[g](n, seed=seed) for g, n, seed in parts)


def main_large():
    ref = Ref()
    blobs, meta = {}, []
    for name, parts, level, checksum in LARGE:
        data = large_plain(parts)
        comp = ref.compress(data, level=level, checksum=checksum)
        blobs[name] = np.frombuffer(comp, dtype=np.uint8)
        meta.append({"key": name, "parts": parts, "n": len(data), "level": level, "checksum": checksum, "csize": len(comp), "sha256": sha256(data)})
    blobs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, "large_frames.npz"), **blobs)
    print(len(meta), "large frames,", sum(m["csize"] for m in meta), "compressed bytes")


if __name__ == "__main__":
    main()
    main_large()
