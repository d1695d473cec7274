"""Dev: compress one saved input several times with the saved options and say where the outputs differ."""
import json, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import warnings
import zstdlite_amd as z
x = open(sys.argv[1], "rb").read()
opts = json.load(open(sys.argv[1].replace(".bin", ".json")))
opts.pop("dict", None)
warnings.simplefilter("ignore")
outs = [z.zstd_compress(x, **opts) for _ in range(8)]
base = outs[0]
for i, o in enumerate(outs[1:], 1):
    if o != base:
        k = next((j for j in range(min(len(o), len(base))) if o[j] != base[j]), min(len(o), len(base)))
        print(f"run {i}: differs from run 0 at byte {k} of {len(base)} / {len(o)}")
print("distinct outputs:", len(set(outs)), "options", opts, "n", len(x))
for lvl in (3, 5):
    o2 = dict(opts, level=lvl)
    print("level", lvl, "distinct:", len({z.zstd_compress(x, **o2) for _ in range(6)}))
o3 = dict(opts); o3.pop("include_checksum", None)
print("no checksum distinct:", len({z.zstd_compress(x, **o3) for _ in range(6)}))
o4 = dict(opts, frame_layout="frames")
print("frames layout distinct:", len({z.zstd_compress(x, **o4) for _ in range(6)}))
