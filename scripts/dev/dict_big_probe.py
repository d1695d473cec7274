"""Dev: large frames the REFERENCE wrote with a dictionary: which route do they take here, how fast?"""
import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, Ref, Oracle
ref = Ref(); o = Oracle()
t_ = GENERATORS["text"](1_300_000, seed=77)
D = ref.train_dict([t_[i * 3000:(i + 1) * 3000] for i in range(400)], 20000)
print("dictionary", len(D), "bytes, formatted:", D[:4] == bytes([0x37, 0xA4, 0x30, 0xEC]))
for mib, lvl in ((4, 3), (32, 3), (32, 1), (32, 6)):
    data = GENERATORS["text"](mib << 20, seed=5)
    frame = ref.compress(data, level=lvl, dict=D)
    seqs, nlit = o.first_block_sequences(frame) if False else (None, None)
    dctx = z.zstd_dctx(dict=D)
    for rep in range(2):
        t0 = time.time(); back = z.zstd_decompress(frame, dctx=dctx); dt = time.time() - t0
    print(f"{mib} MiB level {lvl}: frame {len(frame)} bytes, decoded ok {back == data}, large frames taken {int(_lib.lib().zl_dctx_large_frames(dctx._ptr))}, {len(data) / dt / 1e9:.3f} GB/s")
