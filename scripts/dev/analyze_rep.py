"""How many sequences of our frames could use repeat-offset codes 2/3 (and 1 with the full rules)?"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zl_testlib import GENERATORS, Oracle
o = Oracle()
for gen in ("text", "iid10"):
    data = GENERATORS[gen](4 << 20)
    comp = z.zstd_compress(data)
    pos = 0; tot = 0; r1h = 0; r2h = 0; r3h = 0; r1m1 = 0; expl_bits = 0
    while pos < len(comp):
        fs = o.L.zlo_find_frame_compressed_size(comp[pos:], len(comp) - pos)
        seqs, nlit = o.first_block_sequences(comp[pos:pos + fs])
        r = [1, 4, 8]
        for ll, ml, off in seqs:
            tot += 1
            if ll > 0:
                if off == r[0]: r1h += 1
                elif off == r[1]: r2h += 1; r = [r[1], r[0], r[2]]
                elif off == r[2]: r3h += 1; r = [r[2], r[0], r[1]]
                else: expl_bits += int(off + 3).bit_length(); r = [off, r[0], r[1]]
            else:
                if off == r[1]: r2h += 1; r = [r[1], r[0], r[2]]
                elif off == r[2]: r3h += 1; r = [r[2], r[0], r[1]]
                elif off == r[0] - 1 and off: r1m1 += 1; r = [off, r[0], r[1]]
                else: expl_bits += int(off + 3).bit_length(); r = [off, r[0], r[1]]
        pos += fs
    print(gen, "first-block sequences", tot, "rep1", r1h, "rep2", r2h, "rep3", r3h, "rep1-1", r1m1, "compressed", len(comp))
