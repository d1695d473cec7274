import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zl_testlib import GENERATORS, Oracle, Ref
o = Oracle()
for gen in ("text",):
    data = GENERATORS[gen](4 << 20)
    comp = z.zstd_compress(data)
    pos = 0; tot = 0; cont = 0; ml_hist = np.zeros(300, dtype=np.int64); rep2=0; rep3=0
    while pos < len(comp):
        fs = o.L.zlo_find_frame_compressed_size(comp[pos:], len(comp) - pos)
        seqs, nlit = o.first_block_sequences(comp[pos:pos + fs])
        prev = None; r=[1,4,8]
        for ll, ml, off in seqs:
            tot += 1
            ml_hist[min(int(ml), 299)] += 1
            if ll == 0 and prev is not None and off == prev: cont += 1
            if ll>0 and off!=r[0]:
                if off==r[1]: rep2+=1
                elif off==r[2]: rep3+=1
            if off!=r[0]: r=[off,r[0],r[1]] if off not in r else ([off]+[x for x in r if x!=off])
            prev = off
        pos += fs
    print(gen, "seqs", tot, "continuations (ll==0, same offset)", cont, "ml>=40", int(ml_hist[40:].sum()), "rep2", rep2, "rep3", rep3, "compressed", len(comp))
    rc = Ref().compress(data, level=3)
    seqs, nlit = o.first_block_sequences(rc)
    print("reference first block: seqs", len(seqs), "mean ml", seqs[:,1].mean(), "ours mean ml", (ml_hist*np.arange(300)).sum()/ml_hist.sum())
