"""Dev: the first block of what we write for one of the ratio sweep's inputs: sizes and the first sequences (ll, ml, offset code value)."""
import sys, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zl_testlib import Oracle, Ref
from ratio_inputs import INPUTS
o = Oracle(); ref = Ref()
kind, n = sys.argv[1], int(sys.argv[2])
levels = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3]
d = INPUTS[kind](n)
print("first 64 bytes", d[:64].hex())
for lv in levels:
    for who, out in (("ours", z.zstd_compress(d, level=lv)), ("ref", ref.compress(d, level=lv))):
        seqs, nlit = o.first_block_sequences(out)
        print(who, "level", lv, "size", len(out), "first block: sequences", len(seqs), "literals", nlit, "match bytes", int(seqs[:, 1].sum()) if len(seqs) else 0)
        print("   ", [tuple(int(v) for v in s) for s in seqs[:24]])
