"""Compare our encoder's first block with the reference's on the same 128 KiB input."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import zstdlite_amd as z
from zl_testlib import *

o = Oracle(); r = Ref()

def sections(frame):
    fhd = frame[4]
    single = (fhd >> 5) & 1; fcs = fhd >> 6; did = fhd & 3
    hs = 5 + (0 if single else 1) + (4 if did == 3 else did) + ((1 if single else 0) if fcs == 0 else (1 << fcs))
    bh = int.from_bytes(frame[hs:hs + 3], "little")
    btype, bsize = (bh >> 1) & 3, bh >> 3
    blk = frame[hs + 3: hs + 3 + bsize]
    if btype != 2:
        return {"btype": btype, "bsize": bsize}
    t, sf = blk[0] & 3, (blk[0] >> 2) & 3
    if t < 2:
        lh = [1, 2, 1, 3][sf]; regen = int.from_bytes(blk[:lh], "little") >> (3 if lh == 1 else 4)
        lsec = lh + (regen if t == 0 else 1)
    else:
        lh = [3, 3, 4, 5][sf]; bits = [10, 10, 14, 18][sf]
        h = int.from_bytes(blk[:lh], "little"); regen = (h >> 4) & ((1 << bits) - 1); csz = h >> (4 + bits)
        lsec = lh + csz
    return {"btype": btype, "bsize": bsize, "lit_type": t, "lit_regen": regen, "lit_sec": lsec, "seq_sec": bsize - lsec}

for gen in sys.argv[1:] or ["text"]:
    data = GENERATORS[gen](131072)
    ours = z.zstd_compress(data); theirs = r.compress(data, 3)
    for name, f in (("ours", ours), ("ref ", theirs)):
        s = sections(f)
        seqs, lit = o.first_block_sequences(f)
        ml = seqs[:, 1]; off = seqs[:, 2]; ll = seqs[:, 0]
        rep = int(np.sum(off[1:] == off[:-1]))
        print(gen, name, len(f), s, "nseq", len(seqs), "meanML %.2f" % ml.mean(), "lits", lit, "rep1-like", rep,
              "ml>=40", int((ml >= 40).sum()), "ll0", int((ll == 0).sum()), "medoff", int(np.median(off)))
        print("   bits/seq %.2f  bits/lit %.2f" % (8 * s["seq_sec"] / len(seqs), 8 * s["lit_sec"] / max(lit, 1)))
