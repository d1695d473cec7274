"""Where the bytes go on one input of the small-size ratio sweep (tests/ratio_inputs.py): section sizes of our frame and the
reference's, and the first block's sequences of both.  usage: python scripts/ratio_diag.py name:size [name:size ...]"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import zstdlite_amd as z
from frame_stats import show
from ratio_inputs import INPUTS
from zl_testlib import Oracle, Ref

ref, o = Ref(), Oracle()
for arg in sys.argv[1:]:
    name, n = arg.split(":")
    n = int(n)
    data = INPUTS[name](n)
    print(f"== {name} {n}")
    for who, comp in (("ours", z.zstd_compress(data)), ("ref", ref.compress(data, 3))):
        show(who, comp, n)
        seqs, nlit = o.first_block_sequences(comp)
        if len(seqs):
            print(f"      first block: {len(seqs)} seqs, {nlit} lits;",
                  " LL", collections.Counter(seqs[:, 0].tolist()).most_common(5), " ML", collections.Counter(seqs[:, 1].tolist()).most_common(5),
                  " OF", collections.Counter(seqs[:, 2].tolist()).most_common(5))
            if len(seqs) <= 40:
                print("      ", [tuple(int(x) for x in s) for s in seqs])
