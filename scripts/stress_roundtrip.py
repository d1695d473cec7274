"""Repeated 1 GiB round trips; on a mismatch say whether the encoder or the decoder misbehaved and where."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_mixed, CHUNK
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
warm = len(sys.argv) > 3
src, kinds, q = gen_mixed(n, dev)
if warm:  # mimic the test suite: smaller jobs first, so that the context buffers grow later
    c0 = DeviceCodec()
    s0 = src[: 3 << 20].contiguous()
    c0.compress(s0)
codec = DeviceCodec()
if warm:
    comp0, t0, f0 = codec.compress(src[: 5 << 20].contiguous())
ref_digest = None
bad = 0
for it in range(reps):
    comp, total, fs = codec.compress(src)
    dig = int(comp[:total].to(torch.int64).sum().item())
    if ref_digest is None: ref_digest = (dig, comp[:total].clone(), fs.copy())
    elif dig != ref_digest[0]:
        print(f"iter {it}: ENCODER output differs from iter 0 (sizes equal: {np.array_equal(fs, ref_digest[2])})")
        off = np.concatenate([[0], np.cumsum(fs)])
        if np.array_equal(fs, ref_digest[2]):
            d = (comp[:total] != ref_digest[1]).nonzero().flatten()
            first = int(d[0]); fr = int(np.searchsorted(off, first, side="right") - 1)
            print(f"   {d.numel()} bytes differ, first at {first} = frame {fr} (+{first - int(off[fr])}), kind {kinds[min(3, fr * CHUNK // q)]}")
        bad += 1
    back = torch.empty_like(src)
    tbl = codec.frame_table(fs, codec.chunk_content_sizes(n))
    for dec in range(2):
        back.zero_()
        got, res = codec.decompress(comp, tbl, back)
        if not torch.equal(back, src):
            d = (back != src).nonzero().flatten()
            first = int(d[0]); fr = first // CHUNK
            frames = torch.unique(d // CHUNK)
            print(f"iter {it} decode {dec}: MISMATCH {d.numel()} bytes in {frames.numel()} frames, first at byte {first} = frame {fr} +{first % CHUNK}, kind {kinds[min(3, first // q)]}; frames {frames[:8].tolist()}")
            print("   res ok:", bool((res == CHUNK).all()))
            bad += 1
print("done, problems:", bad)
