"""Compression ratio of the HIP encoder next to the reference's level 3 on the long-range inputs
(zl_testlib.LONG_RANGE) and the SURVEY 8(d) inputs.  Runs on the GPU box: python scripts/ratio_longrange.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import zstdlite_amd as z
from zl_testlib import GENERATORS, LONG_RANGE, Ref, have_ref

ref = Ref() if have_ref() else None
out = {}
levels = [int(x) for x in (sys.argv[1:] or ["3"])]
for name, n in (("corpus", None), ("records", 16 << 20), ("text", 8 << 20), ("iid10", 8 << 20), ("int20", 8 << 20)):
    data = (LONG_RANGE.get(name) or GENERATORS[name])(n)
    for level in levels:
        cctx = z.zstd_cctx(level=level)
        comp = z.zstd_compress(data, cctx=cctx)
        assert z.zstd_decompress(comp) == data
        row = {"bytes": len(data), "ours": len(data) / len(comp)}
        if ref is not None:
            rc = ref.compress(data, level=level)
            assert ref.decompress(comp, len(data)) == data if len(comp) == ref.L.ZSTD_findFrameCompressedSize(comp, len(comp)) else True
            row["ref"] = len(data) / len(rc)
            row["ours_over_ref"] = row["ours"] / row["ref"]
        out[f"{name}@L{level}"] = row
        print(name, level, row, flush=True)
print(json.dumps(out))
