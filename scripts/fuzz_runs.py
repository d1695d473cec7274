"""Fuzz the decode of a frame in RUNS of blocks (zl_api.hip decode_walked_frame_in_runs: how frames of 4 GiB and more that someone
else wrote are decoded) with frames of a few MiB: ZL_d_runBlocks = N sends every frame of more than N blocks through that path.
Reference-made frames (levels 1 .. 19, with and without checksum, compressible, incompressible and mixed so that raw blocks,
treeless literals and Repeat_Mode all occur at run boundaries) and this library's own one frame.  Intact: bit-exact, and the runs
must actually have been taken; damaged or cut short: the verdict and the bytes of the oracle.
Round 4: the host entry also with ZL_d_slabMiB = 1 -- the same runs with the upload, the download and the hash beside them.
usage: fuzz_runs.py [rounds=20] [seed=1]"""
import ctypes as C
import sys
import time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, LONG_RANGE, Oracle, Ref, is_error
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o = Oracle(); L = _lib.lib(); ref = Ref()
noise = rng.integers(0, 256, 3 << 20, dtype=np.uint8).tobytes()
text = GENERATORS["text"](6 << 20, seed=11)
ints = GENERATORS["int20"](4 << 20, seed=12)
low = GENERATORS["iid10"](5 << 20, seed=13)
pieces = []
for k in range(24):  # 256 KiB pieces of alternating kinds: tables come and go between neighbouring blocks
    src = (text, noise, ints, low, bytes(1 << 20))[k % 5]
    at = int(rng.integers(0, len(src) - (256 << 10)))
    pieces.append(src[at:at + (256 << 10)])
mixed = b"".join(pieces)
base = []
for data, kw in ((text, {"level": 3}), (low, {"level": 3, "checksum": True}), (ints + bytes(300000), {"level": 1}), (noise + text[:2 << 20] + noise[:1 << 20], {"level": 3}),
                 (mixed, {"level": 3, "checksum": True}), (mixed, {"level": 9}), (text[:3 << 20], {"level": 19}), (LONG_RANGE["records"](4 << 20), {"level": 5}),
                 (noise, {"level": 3}), (low[:3 << 20], {"level": -5})):
    base.append((ref.compress(data, **kw), data))
base.append((z.zstd_compress(text[:4 << 20] + low[:1 << 20]), text[:4 << 20] + low[:1 << 20]))
print(len(base), "base frames;", sum(len(f) for f, _ in base), "compressed bytes", flush=True)
dctx = L.zl_createDCtx()
bad = 0
t0 = time.time()


import torch
from zstdlite_amd.device import DeviceCodec
codec = DeviceCodec()


def ours_device(f, n, nrun, with_copy):
    """the device-resident entry: with the caller's host copy of the frame, or with device pointers only (the library then
    brings the compressed bytes down itself for the run plan); the content checksum is hashed on the host from device memory"""
    assert L.zl_DCtx_setParameter(codec.dctx, 9009, nrun) == 0
    h = np.frombuffer(f, dtype=np.uint8).copy()
    d_src = torch.from_numpy(h).cuda()
    d_dst = torch.zeros(max(n, 1), dtype=torch.uint8, device="cuda")
    try:
        total, _ = codec.decompress(d_src, codec.frame_table([len(f)], [n]), d_dst, host_copy=h if with_copy else None)
    except z.ZstdError:
        return (1 << 64) - 1, None
    return total, d_dst.cpu().numpy()


def ours(f, n, nrun, slab=0):
    """the host entry; slab: ZL_d_slabMiB = 1 -- the frame moves slab by slab (upload, runs, download and hash side by side)"""
    assert L.zl_DCtx_setParameter(dctx, 9009, nrun) == 0 and L.zl_DCtx_setParameter(dctx, 9010, slab) == 0
    out = np.zeros(max(n, 1), dtype=np.uint8)
    src = np.frombuffer(f, dtype=np.uint8)
    r = L.zl_decompressDCtx(dctx, out.ctypes.data, n, src.ctypes.data, len(f))
    return r, out


# intact frames, every run length: bit-exact, and in runs
for i, (f, data) in enumerate(base):
    took = []
    for nrun, slab in ((16, 0), (17, 0), (24, 0), (33, 0), (64, 0), (0, 1), (16, 1), (33, 1)):
        before = L.zl_dctx_frame_runs(dctx)
        r, out = ours(f, len(data), nrun, slab)
        if is_error(r) or r != len(data) or out[:r].tobytes() != data:
            bad += 1
            print(f"FAIL intact frame {i} run length {nrun}: {_lib.error_name(r) if is_error(r) else r}")
        took.append(int(L.zl_dctx_frame_runs(dctx) - before))
    print(f"frame {i}: {len(data) >> 10} KiB -> {len(f) >> 10} KiB, runs taken per run length {took}", flush=True)
accepted = rejected = 0
for rd in range(rounds):
    for i, (f, data) in enumerate(base):
        b = bytearray(f)
        kind = int(rng.integers(0, 6))
        if kind == 5:
            b = b[: int(rng.integers(8, len(b)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(4, len(b)))
                b[pos] ^= 1 << int(rng.integers(0, 8))
        b = bytes(b)
        nrun = int(rng.integers(16, 80))
        code, want = o.decompress_code(b, len(data))
        entry = int(rng.integers(0, 4))  # host entry twice as often as each flavour of the device entry
        if entry < 2 or len(b) < 32 or is_error(o.L.zlo_find_frame_compressed_size(b, len(b))) or o.L.zlo_find_frame_compressed_size(b, len(b)) != len(b):
            r, out = ours(b, len(data), nrun, int(rng.integers(0, 2)))  # (the device entry is handed exactly one frame: the host entry takes the cut and the glued ones)
        else:
            r, out = ours_device(b, len(data), nrun, entry == 2)
        if is_error(code) != is_error(r):
            bad += 1
            print(f"FAIL round {rd} frame {i} run length {nrun}: oracle {o.error_name(code) if is_error(code) else code}, ours {_lib.error_name(r) if is_error(r) else r}")
        elif not is_error(r):
            accepted += 1
            if r != code or out[:r].tobytes() != want:
                bad += 1
                print(f"FAIL round {rd} frame {i} run length {nrun}: accepted with other bytes")
        else:
            rejected += 1
print(f"{rounds} rounds, {accepted} accepted, {rejected} rejected, failures {bad}, {time.time() - t0:.1f}s")
sys.exit(1 if bad else 0)
