"""Compressed size against the reference's level 3 where R objects live: 1 KB to 1 MB (the 5 % bar was only
pinned at 8 MiB and above).  Inputs: the SURVEY 8(d) generators, float64 vectors (R's default numeric: what man/benchmark.R's
data frame holds), the hand-built serialization of head(mtcars) (config C1) and a pickled data frame.  Prints a table and
writes it as JSON (profiles/r03_ratio_sweep.json is the committed copy; tests/test_compress_gpu.py reads the same inputs).
usage: python scripts/ratio_sweep.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import zstdlite_amd as z
from zl_testlib import Ref
from ratio_inputs import SIZES, INPUTS

ref = Ref()
table = {}
print(f"{'input':14s}" + "".join(f"{s:>10d}" for s in SIZES))
for name, gen in INPUTS.items():
    row = {}
    for n in SIZES:
        data = gen(n)
        ours = z.zstd_compress(data)
        assert ref.decompress(ours, len(data)) == data
        theirs = ref.compress(data, level=3)
        row[str(n)] = {"bytes": len(data), "ours": len(ours), "reference": len(theirs), "size_ratio": len(ours) / len(theirs)}
    table[name] = row
    print(f"{name:14s}" + "".join(f"{row[str(s)]['size_ratio']:10.3f}" for s in SIZES), flush=True)
if len(sys.argv) > 1:
    json.dump(table, open(sys.argv[1], "w"), indent=1)
