"""Dev: i.i.d. bytes over ten symbols -- ratio by level and layout, and what the first block of a frame holds."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import GENERATORS, Oracle
o = Oracle()
for n in (131072, 6 << 20):
    d = GENERATORS["iid10"](n, seed=5)
    for lv in (1, 3, 6):
        out = z.zstd_compress(d, level=lv, frame_layout="frames")
        first = z.zstd_info(out)["compressed_size"]
        seqs, nlit = o.first_block_sequences(out)
        lens = [int(s[1]) for s in seqs]; offs = [int(s[2]) for s in seqs]
        print(n, lv, round(n / len(out), 4), "first frame", first, "bytes; sequences", len(seqs), "match bytes", sum(lens), "min/max", min(lens, default=0), max(lens, default=0), "short (<8)", sum(1 for l in lens if l < 8), "literals", nlit, "offsets<=3", sum(1 for f in offs if f <= 3))
