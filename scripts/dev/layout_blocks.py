"""Dev: the first block of the same text in the two layouts: sequences and literals (why the one-frame layout's entropy stage is slower)."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import GENERATORS, Oracle
o = Oracle()
for kind in ("text", "iid10", "int20"):
    d = GENERATORS[kind](32 << 20, seed=5)   # (above 16 MiB: the table parse in both)
    for lay in ("frames", "auto"):
        out = z.zstd_compress(d, frame_layout=lay)
        seqs, nlit = o.first_block_sequences(out)
        ml = seqs[:, 1].sum() if len(seqs) else 0
        print(f"{kind:6s} {lay:6s}: total {len(out):9d} bytes; first block: {len(seqs):5d} sequences, {nlit:6d} literals, {int(ml):6d} match bytes, block content {int(ml) + nlit}")
