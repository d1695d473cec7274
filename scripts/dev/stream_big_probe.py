"""The streaming decoder's route for ONE large frame (decoded while it arrives), small: frames of a few MiB with ZL_d_slabMiB = 1,
fed in pieces; prints what came out.  usage: stream_big_probe.py"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib, stream
from zl_testlib import GENERATORS, Ref
L = _lib.lib()
ref = Ref()
for kind in ("text", "rand", "iid10", "int20"):
    data = GENERATORS[kind](3000000, seed=81)
    for who, f in (("ref", ref.compress(data, level=3, checksum=True)), ("own", z.zstd_compress(data, include_checksum=True))):
        for piece in (1 << 20, 131702, 4096):
            dctx = z.zstd_dctx()
            L.zl_DCtx_setParameter(dctx._ptr, 9010, 1)
            dec = stream.Decompressor(dctx._ptr)
            got = []
            try:
                for pos in range(0, len(f), piece):
                    got.append(dec.feed(f[pos:pos + piece]))
                ok = b"".join(got) == data and dec.at_frame_boundary
                print(kind, who, piece, "ok" if ok else f"MISMATCH ({sum(map(len, got))} bytes, boundary {dec.at_frame_boundary})", "runs", L.zl_dctx_frame_runs(dctx._ptr), flush=True)
            except Exception as e:
                print(kind, who, piece, "ERROR", e, flush=True)
