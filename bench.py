#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X:
    "GB/s zstd_compress+zstd_decompress on 1 GiB raw, level 3; ratio vs CPU".

One step = one pass of the hot path over one synthetic 1 GiB raw vector already resident in
HBM: zstd_compress (level 3, HIP encoder) followed by zstd_decompress of its output (HIP
decoder), both through the device entry points of the C-ABI.  value = uncompressed bytes of all
ranks / max-over-ranks step time (GB = 1e9 B).  The vector is one quarter each of the four
SURVEY 8(d) inputs (rand / iid10 / text / int20), so the reported compression ratio can be read
against BASELINE.md section 2, which quotes the reference's level-3 ratio on 256 MiB of each.

N > 1 (launched by torch.distributed.run, one rank per GPU): frames are independent, so every
rank compresses and decompresses its own 1 GiB shard with no data-path collective ("weak"
scaling); the one exchange step the path has -- collecting the compressed shards on rank 0, sizes
by all-gather then payloads by grouped send/recv over RCCL -- is inside the timed step.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed on the launch
stream), "cpu_baseline" (the reference's bundled libzstd, oracle/_ref, on the host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def _cpu_info():
    model, smt = "unknown", None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        smt = open("/sys/devices/system/cpu/smt/active").read().strip() == "1"
    except OSError:
        pass
    return model, smt


def cpu_baseline(host_bytes, log, frames=None, reps=5):
    """The reference's CPU path on this box's host cores: ZSTD_compress2(level 3) and
    ZSTD_decompressDCtx on the same 1 GiB vector, buffers pre-touched, best of `reps` passes: (i) one thread,
    (ii) compression with nbWorkers = host cores, (iii) decode of a set of independent frames (`frames`: the HIP
    encoder's per-chunk frames of the same vector) with one thread per core, each calling ZSTD_decompressDCtx on
    its own frames -- libzstd has no multi-threaded decoder.  TEST/BENCH infrastructure (oracle/_ref), never on
    the product path."""
    import threading
    from zl_testlib import REF_SO, have_ref
    if not have_ref():
        return None
    L = C.CDLL(REF_SO)
    L.ZSTD_compressBound.restype = C.c_size_t
    L.ZSTD_compressBound.argtypes = [C.c_size_t]
    L.ZSTD_createCCtx.restype = C.c_void_p
    L.ZSTD_createDCtx.restype = C.c_void_p
    L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ZSTD_CCtx_setParameter.restype = C.c_size_t
    L.ZSTD_compress2.restype = C.c_size_t
    L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_decompressDCtx.restype = C.c_size_t
    L.ZSTD_decompressDCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    L.ZSTD_freeDCtx.argtypes = [C.c_void_p]
    L.ZSTD_getErrorName.restype = C.c_char_p
    L.ZSTD_getErrorName.argtypes = [C.c_size_t]
    n = host_bytes.size
    cap = L.ZSTD_compressBound(n)
    comp = np.empty(cap, dtype=np.uint8)
    back = np.empty(n, dtype=np.uint8)
    comp.fill(1)  # pre-touch: first-touch page faults otherwise dominate
    back.fill(1)
    res = {}
    ncores = os.cpu_count() or 1
    for label, workers in (("mt", ncores), ("1t", 0)):  # 1-thread output stays in the buffer for the decode leg
        best = 1e9
        for _ in range(reps):
            cctx = L.ZSTD_createCCtx()
            L.ZSTD_CCtx_setParameter(cctx, 100, 3)
            if workers:
                L.ZSTD_CCtx_setParameter(cctx, 400, workers)
            t0 = time.perf_counter()
            csize = L.ZSTD_compress2(cctx, comp.ctypes.data, cap, host_bytes.ctypes.data, n)
            best = min(best, time.perf_counter() - t0)
            L.ZSTD_freeCCtx(cctx)
        res[label] = (best, csize)
        log(f"cpu libzstd compress {label}: {n / best / 1e9:.3f} GB/s, {csize} bytes")
    tc, csize = res["1t"]
    td = 1e9
    for _ in range(reps):
        dctx = L.ZSTD_createDCtx()
        t0 = time.perf_counter()
        r = L.ZSTD_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, csize)
        td = min(td, time.perf_counter() - t0)
        L.ZSTD_freeDCtx(dctx)
    assert r == n, (r, L.ZSTD_getErrorName(r))
    assert np.array_equal(back, host_bytes)
    ref_frame = comp[:csize].copy()
    mt_decode = None
    if frames is not None:  # (iii): frames = (blob, sizes, content sizes); thread k takes every ncores-th run of frames
        blob, fs, cs = frames
        src_off = np.concatenate([[0], np.cumsum(fs)[:-1]]).astype(np.int64)
        dst_off = np.concatenate([[0], np.cumsum(cs)[:-1]]).astype(np.int64)
        nthreads = min(ncores, len(fs))
        back.fill(0)

        def work(k):
            dctx = L.ZSTD_createDCtx()
            per = -(-len(fs) // nthreads)
            for i in range(k * per, min(len(fs), (k + 1) * per)):
                L.ZSTD_decompressDCtx(dctx, back.ctypes.data + int(dst_off[i]), int(cs[i]), blob.ctypes.data + int(src_off[i]), int(fs[i]))
            L.ZSTD_freeDCtx(dctx)
        best = 1e9
        for _ in range(reps):
            ths = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            best = min(best, time.perf_counter() - t0)
        assert np.array_equal(back, host_bytes)
        mt_decode = {"GBps": n / best / 1e9, "threads": nthreads, "frames": int(len(fs))}
        log(f"cpu libzstd decode of {len(fs)} frames on {nthreads} threads: {n / best / 1e9:.2f} GB/s")
    model, smt = _cpu_info()
    return {
        "value": n / (tc + td) / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference",
        "sample": f"the full {n >> 20} MiB bench vector, best of {reps}: ZSTD_compress2(level 3) + ZSTD_decompressDCtx, bundled libzstd 1.5.6 built -O2 from the reference source (oracle/_ref), 1 thread",
        "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9, "ratio": n / csize, "compressed_bytes": int(csize),
        "compress_GBps_nbWorkers": n / res["mt"][0] / 1e9, "nbWorkers": ncores, "host_cores": ncores,
        "decompress_frames_threads": mt_decode, "cpu_model": model, "smt_active": smt, "best_of": reps,
        "_frame": ref_frame,  # (the reference's single frame of the whole vector: input of the N1 leg, removed before printing)
    }


def one_frame_leg(device, src, log):
    """The windowed layout (include/zl_hip.h ZL_c_frameLayout: ONE frame, matches up to 2 MiB back across chunks: what
    zl_compress2 writes by default), device-resident: compress, then decode with the chunk hint."""
    import torch
    from zstdlite_amd.device import DeviceCodec
    n = int(src.numel())
    codec = DeviceCodec(level=3, profile=True, frame_layout="one_frame")
    comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=device)
    back = torch.empty(n, dtype=torch.uint8, device=device)
    tc = td = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, total, fs = codec.compress(src, comp)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        back.zero_()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        got, _ = codec.decompress(comp, codec.frame_table([total], [n]), back, chunk_sizes=fs)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if it:
            tc, td = min(tc, t1 - t0), min(td, t3 - t2)
    assert got == n and torch.equal(back, src), "one-frame round trip is not bit-exact"
    km = codec.kernel_ms()
    codec.close()
    ckm = {k: km[k] for k in ("far", "match", "entropy", "scan", "gather") if k in km}
    # roofline of this layout's dominant kernel, as the headline's: algorithmic bytes (input read once + frame written once) over
    # the kernel's HIP-event duration ("far" = the two far-table passes together; the PMC traffic of these kernels is under
    # profiles/, taken by scripts/profile_round.sh with the same one-frame round trip)
    dom = max(ckm, key=ckm.get)
    alg = n + total
    ach = alg / (ckm[dom] / 1e3) / 1e9
    return {"ratio": n / total, "compressed_bytes": int(total), "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9,
            "compress_kernel_ms": ckm, "bit_exact": True,
            "roofline": {"bound": "hbm", "kernel": {"match": "zl_match_auto_kernel<false, false, true, true, false>", "far": "zl_far_build_kernel + zl_far_probe_kernel",
                                                    "entropy": "zl_entropy_kernel", "gather": "zl_gather_kernel", "scan": "zl_scan_sizes_kernel"}[dom],
                         "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": ckm[dom]}}


def per_input_full_leg(codec, device, n, log):
    """Every SURVEY 8(d) input at the FULL size on its own -- BASELINE configs[2] read literally is the first row: a 1 GiB
    vector of uniform random bytes (8192 raw-block frames) -- through the same device entries as the headline: round trip
    checked bit for bit, best of three."""
    import torch
    from zstdlite_amd.device import gen_device
    out = {}
    comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=device)
    back = torch.empty(n, dtype=torch.uint8, device=device)
    content = codec.chunk_content_sizes(n)
    for kind in ("rand", "iid10", "text", "int20"):
        src = gen_device(kind, n, device)
        tc = td = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, total, fs = codec.compress(src, comp)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            got, _ = codec.decompress(comp, codec.frame_table(fs, content), back)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            tc, td = min(tc, t1 - t0), min(td, t2 - t1)
        assert got == n and torch.equal(back, src), f"{kind}: round trip is not bit-exact"
        assert total <= codec.bound(n)
        km = codec.kernel_ms()
        out[kind] = {"bytes": n, "compressed_bytes": int(total), "ratio": n / total, "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9,
                     "roundtrip_GBps": n / (tc + td) / 1e9, "match_ms": km.get("match"), "bit_exact": True}
        log(f"per input, full size {kind}: {out[kind]}")
        del src
    return out


def single_object_leg(log):
    """ONE object from host memory to host memory through the C entries alone (no Python in the timed region, output buffer made in
    front), 64 KiB - 16 MiB, where zstd_serialize() of ordinary R objects lives: a call there is bound by its launches and by the
    chain of ONE block, not by the device's throughput.  Wall time of zl_compress2 / zl_decompressDCtx (level 3, text; this library's
    own frame and the reference's), next to the reference on one host core.  Round 5 found two cliffs here: the
    table makes them visible."""
    import zstdlite_amd as z
    from zstdlite_amd import _lib
    from zl_testlib import GENERATORS, Ref, have_ref
    ref = Ref() if have_ref() else None
    L = _lib.lib()
    cctx, dctx = z.zstd_cctx(), z.zstd_dctx()
    def best(f, reps=4):
        b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t0)
        return b, r
    out = {}
    for n in (64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20):
        data = GENERATORS["text"](n, seed=5)
        cap = L.zl_compressBound(n)
        comp = (C.c_char * cap)()
        back = (C.c_char * n)()
        tc, csz = best(lambda: L.zl_compress2(cctx._ptr, comp, cap, data, n))
        own = bytes(comp[:csz])
        td, r = best(lambda: L.zl_decompressDCtx(dctx._ptr, back, n, own, len(own)))
        assert r == n and bytes(back) == data
        row = {"compress_ms": tc * 1e3, "decompress_own_frame_ms": td * 1e3, "ratio": n / csz}
        if ref is not None:
            trc, rframe = best(lambda: ref.compress(data, level=3), 2)
            trd, _ = best(lambda: ref.decompress(rframe, n), 2)
            tdr, r = best(lambda: L.zl_decompressDCtx(dctx._ptr, back, n, rframe, len(rframe)))
            assert r == n and bytes(back) == data
            row.update({"decompress_reference_frame_ms": tdr * 1e3, "reference_one_core": {"compress_ms": trc * 1e3, "decompress_ms": trd * 1e3, "ratio": n / len(rframe)}})
        out[str(n)] = row
        log(f"one object of {n >> 10} KiB host to host: compress {row['compress_ms']:.2f} ms, decompress {row['decompress_own_frame_ms']:.2f} ms"
            + (f" (the reference's frame: {row['decompress_reference_frame_ms']:.2f}); reference on one core {row['reference_one_core']['compress_ms']:.2f} / {row['reference_one_core']['decompress_ms']:.2f} ms" if ref is not None else ""))
    return out


def levels_leg(device, src, log):
    """the operating points of zstd_cctx(level =) (R/cctx.R:10-14) on the bench vector, device-resident, ONE frame
    (the layout the reference's own zstd_compress() writes): compress rate, stage times (search = the far passes at levels <= 5, the
    deep match finder's sort + search from 6 on) and ratio; next to it the REFERENCE's ratio at the same level on the python
    sources and on 16 MiB of text (host entry; both decoders check the bytes)."""
    import warnings
    import torch
    import zstdlite_amd as z
    from zstdlite_amd.device import DeviceCodec
    from zl_testlib import GENERATORS, LONG_RANGE, Ref, have_ref
    ref = Ref() if have_ref() else None
    n = int(src.numel())
    samples = {"corpus": LONG_RANGE["corpus"](None), "text16M": GENERATORS["text"](16 << 20)}
    out = {}
    for level in (-5, 1, 3, 4, 5, 6, 9, 12, 15, 16, 19):  # (16, 19: the price-based parse of csrc/zl_opt.hip, round 6; parse_ms then holds the lazy parse AND its passes)
        codec = DeviceCodec(level=level, profile=True, frame_layout="one_frame")
        dst = torch.empty(codec.bound(n), dtype=torch.uint8, device=device)
        best = 1e9
        for _ in range(3 if level < 16 else 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _, total, _fs = codec.compress(src, dst)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        km = codec.kernel_ms()
        codec.close()
        row = {"compress_GBps": n / best / 1e9, "ratio": n / total, "search_ms": km.get("far"), "parse_ms": km.get("match"), "entropy_ms": km.get("entropy")}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for name, data in samples.items():
                comp = z.zstd_compress(data, level=level)
                assert z.zstd_decompress(comp) == data
                row[name] = {"ratio": len(data) / len(comp)}
                if ref is not None:
                    assert ref.decompress(comp, len(data)) == data
                    theirs = len(ref.compress(data, level=level))
                    row[name].update(ratio_reference_same_level=len(data) / theirs, compressed_size_ours_over_cpu=len(comp) / theirs)
        out[str(level)] = row
        log(f"level {level}: {row}")
        del dst
    return out


def long_range_leg(device, log):
    """Ratio on inputs whose redundancy lies further back than one 128 KiB chunk (tests/zl_testlib.LONG_RANGE), through the
    host entry as the reference's glue calls it, next to the reference's level 3; both decoders check the bytes."""
    import zstdlite_amd as z
    from zl_testlib import LONG_RANGE, Ref, have_ref
    ref = Ref() if have_ref() else None
    out = {}
    for name, nbytes in (("records", 64 << 20), ("corpus", None)):
        data = LONG_RANGE[name](nbytes)
        comp = z.zstd_compress(data)
        frames = z.zstd_compress(data, frame_layout="frames")
        assert z.zstd_decompress(comp) == data
        row = {"bytes": len(data), "ratio_one_frame": len(data) / len(comp), "ratio_independent_frames": len(data) / len(frames)}
        if ref is not None:
            theirs = ref.compress(data, level=3)
            assert ref.decompress(comp, len(data)) == data
            row["ratio_reference_level3"] = len(data) / len(theirs)
            row["compressed_size_ours_over_cpu"] = len(comp) / len(theirs)
        out[name] = row
        log(f"long range {name}: {row}")
    return out


def c4_items(nitems, seed=3):
    """Config C4's objects: pickled dicts of 10-39 of 50 names -> small ints (the README / vignette dictionary example's shape,
    README.md:283-293), about 430 bytes each."""
    import pickle
    rng = np.random.default_rng(seed)
    names = [f"name_{i:02d}_{'x' * (i % 7)}" for i in range(50)]
    items = []
    for _ in range(nitems):
        perm = rng.permutation(50)[: int(rng.integers(10, 40))]
        items.append(pickle.dumps({names[j]: int(rng.integers(0, 1000)) for j in perm}, protocol=4))
    return items


class C4Batch:
    """The objects packed ONCE into one buffer with pointer / size arrays, outputs preallocated: what a C caller of
    zl_compress_batch / zl_decompress_batch holds.  No Python in the timed calls."""

    def __init__(self, items, bound):
        self.n = n = len(items)
        self.sizes = np.array([len(x) for x in items], dtype=np.uint64)
        self.off = np.concatenate([[0], np.cumsum(self.sizes)[:-1]]).astype(np.uint64)
        self.blob = np.frombuffer(b"".join(items), dtype=np.uint8).copy()
        self.caps = np.array([bound(int(k)) for k in self.sizes], dtype=np.uint64)
        self.coff = np.concatenate([[0], np.cumsum(self.caps)[:-1]]).astype(np.uint64)
        self.comp = np.zeros(int(self.caps.sum()), dtype=np.uint8)
        self.csize = np.zeros(n, dtype=np.uint64)
        self.back = np.zeros(self.blob.size, dtype=np.uint8)
        self.bsize = np.zeros(n, dtype=np.uint64)
        self.p_src = (self.blob.ctypes.data + self.off).astype(np.uint64)
        self.p_comp = (self.comp.ctypes.data + self.coff).astype(np.uint64)
        self.p_back = (self.back.ctypes.data + self.off).astype(np.uint64)

    @staticmethod
    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)


def c4_leg(log, nitems=100000):
    """BASELINE config C4: zstd_serialize() of 100k small list elements with a trained dictionary (ZDICT_trainFromBuffer, size
    5000, over the first 2000 objects, by the reference library).  Three timings: (i) the C-ABI batch entries on arrays packed
    once (no Python in the call), with the call's phases from the library; (ii) the Python mirror's zstd_compress_batch (what a
    script pays: a bytes object per item); (iii) the reference's per-object loop on one thread and on every host core, each
    thread with its own context and dictionary (oracle/zl_ref_batch.c)."""
    import zstdlite_amd as z
    from zstdlite_amd import _lib
    from zl_testlib import REF_SO, Ref, have_ref
    if not have_ref():
        return None
    ref = Ref()
    L = _lib.lib()
    t0 = time.perf_counter()
    items = c4_items(nitems)
    d = ref.train_dict(items[:2000], 5000)
    raw = sum(map(len, items))
    log(f"C4: {nitems} objects prepared in {time.perf_counter() - t0:.1f}s")
    B = C4Batch(items, L.zl_compressBound)
    vp = C.POINTER(C.c_void_p)
    sp = C.POINTER(C.c_size_t)
    for fn in ("zl_cctx_batch_ms", "zl_dctx_batch_ms"):
        getattr(L, fn).argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        getattr(L, fn).restype = C.c_size_t
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_loadDictionary(cctx, d, len(d)); L.zl_DCtx_loadDictionary(dctx, d, len(d))
    # The calls are timed as a caller makes them (a large batch is packed, uploaded, brought down and unpacked in pieces, round 6); ZL_c_profile /
    # ZL_d_profile = 1 drains the stream between the phases of a call to time them, so the phases come from calls of their own behind the timed ones
    tc = td = 1e9
    cms, dms = (C.c_double * 5)(), (C.c_double * 5)()
    # (the legs in front have given back tens of GiB of device memory, which the driver wipes with the copy engines: copies that start right behind
    # it run at half their rate for some hundred milliseconds -- DESIGN section 6; a call of 4 ms is all copies and launches)
    time.sleep(1.0)
    for it in range(14):
        prof = 1 if it >= 12 else 0
        L.zl_CCtx_setParameter(cctx, 9001, prof); L.zl_DCtx_setParameter(dctx, 9002, prof)
        t1 = time.perf_counter()
        r = L.zl_compress_batch(cctx, B.n, B.p_comp.ctypes.data_as(vp), B.caps.ctypes.data_as(sp), B.p_src.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.csize.ctypes.data_as(sp))
        t2 = time.perf_counter()
        assert not _lib.is_error(r), _lib.error_name(r)
        r = L.zl_decompress_batch(dctx, B.n, B.p_back.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.p_comp.ctypes.data_as(vp), B.csize.ctypes.data_as(sp), B.bsize.ctypes.data_as(sp))
        t3 = time.perf_counter()
        assert not _lib.is_error(r), _lib.error_name(r)
        if prof:
            L.zl_cctx_batch_ms(cctx, cms); L.zl_dctx_batch_ms(dctx, dms)
        elif it:
            tc, td = min(tc, t2 - t1), min(td, t3 - t2)
    assert (B.bsize == B.sizes).all() and np.array_equal(B.back, B.blob), "C4 batch round trip is not bit-exact"
    ours_total = int(B.csize.sum())
    for i in range(0, nitems, 50):  # one frame in fifty through the reference decoder with the dictionary
        f = B.comp[int(B.coff[i]):int(B.coff[i]) + int(B.csize[i])].tobytes()
        assert ref.decompress(f, len(items[i]), dict=d) == items[i]
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
    # (ii) the Python mirror
    cctx_py = z.zstd_cctx(dict=d)
    z.zstd_compress_batch(items[:1000], cctx=cctx_py)
    t1 = time.perf_counter(); ours_py = z.zstd_compress_batch(items, cctx=cctx_py); t_py_c = time.perf_counter() - t1
    t1 = time.perf_counter(); back_py = z.zstd_decompress_batch(ours_py, dict=d); t_py_d = time.perf_counter() - t1
    assert back_py == items
    # (iii) the reference: one thread, every core
    R = C.CDLL(os.path.join(os.path.dirname(REF_SO), "libzl_ref_batch.so"))
    R.zlr_init.argtypes = [C.c_char_p]
    R.zlr_batch.restype = C.c_double
    R.zlr_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t] + [C.c_void_p] * 7
    assert R.zlr_init(REF_SO.encode()) == 0
    ncores = os.cpu_count() or 1
    rcomp = np.zeros_like(B.comp); rsize = np.zeros(B.n, dtype=np.uint64)
    rback = np.zeros_like(B.back); rbsize = np.zeros(B.n, dtype=np.uint64)
    cpu = {}
    for label, th in (("1_thread", 1), ("all_cores", ncores)):
        bc = bd = 1e9
        for _ in range(3):
            bc = min(bc, R.zlr_batch(0, 3, th, d, len(d), B.n, B.blob.ctypes.data, B.off.ctypes.data, B.sizes.ctypes.data, rcomp.ctypes.data, B.coff.ctypes.data, B.caps.ctypes.data, rsize.ctypes.data))
            bd = min(bd, R.zlr_batch(1, 3, th, d, len(d), B.n, rcomp.ctypes.data, B.coff.ctypes.data, rsize.ctypes.data, rback.ctypes.data, B.off.ctypes.data, B.sizes.ctypes.data, rbsize.ctypes.data))
        cpu[label] = {"threads": th, "compress_objects_per_s": B.n / bc, "decompress_objects_per_s": B.n / bd}
    assert np.array_equal(rback, B.blob)
    gpu_c, cpu_c = nitems / tc, cpu["all_cores"]["compress_objects_per_s"]
    gpu_d, cpu_d = nitems / td, cpu["all_cores"]["decompress_objects_per_s"]
    return {"objects": nitems, "mean_bytes": raw / nitems, "dict_bytes": len(d), "ratio": raw / ours_total,
            "ratio_reference_same_dict": raw / int(rsize.sum()),
            "c_abi_objects_per_s": {"compress": nitems / tc, "decompress": nitems / td},
            "c_abi_payload_MBps": {"compress": raw / tc / 1e6, "decompress": raw / td / 1e6},
            "kernel_ms": {"compress": cms[2], "decompress": dms[2]},
            "call_phases_ms": {"compress": dict(zip(("pack", "upload", "kernels", "download", "unpack"), (round(x, 3) for x in cms))),
                               "decompress": dict(zip(("pack", "upload", "kernels", "download", "unpack"), (round(x, 3) for x in dms)))},
            "python_mirror_objects_per_s": {"compress": nitems / t_py_c, "decompress": nitems / t_py_d},
            "cpu_threads_objects_per_s": cpu, "host_cores": ncores,
            "gpu_over_cpu_all_cores_compress": gpu_c / cpu_c,
            "gpu_over_cpu_all_cores_decompress": gpu_d / cpu_d,
            "note": "; ".join(
                (f"{what}: the batch entry on one GPU runs at {g / c:.2f}x the reference's per-object loop on {ncores} host threads" +
                 ("" if g >= c else " (objects of 400 bytes: the call's host side -- a memcpy and a header parse per object -- is half of the call)"))
                for what, g, c in (("compress", gpu_c, cpu_c), ("decompress", gpu_d, cpu_d))),
            "checked": "every frame through our decoder; one in fifty through the reference decoder with the dictionary"}


def end_to_end_leg(host, device, log):
    """What the R glue's call costs: zl_compress2 / zl_decompressDCtx with HOST buffers (src/raw-file.c:74,189), PCIe both
    ways included, next to the PCIe rate of this box measured the same way (pageable host memory)."""
    import torch
    from zstdlite_amd import _lib
    L = _lib.lib()
    n = host.size
    cap = L.zl_compressBound(n)
    comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
    back = np.empty(n, dtype=np.uint8); back.fill(1)
    out = {}
    for layout, code in (("one_frame", 0), ("independent_frames", 1)):
        cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
        L.zl_CCtx_setParameter(cctx, 9003, code)
        tc = td = 1e9
        for _ in range(5):  # (best of 5, as the CPU baseline: the first calls of a context create its streams, peers and buffers)
            t0 = time.perf_counter(); r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n); t1 = time.perf_counter()
            assert not _lib.is_error(r), _lib.error_name(r)
            r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
            assert r2 == n, _lib.error_name(r2)
            tc, td = min(tc, t1 - t0), min(td, t2 - t1)
        assert np.array_equal(back, host)
        L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
        # (the contexts give back ten GiB of device memory, which the driver wipes with the copy engines: downloads that start
        # right behind it ran at half their rate for a few hundred ms -- scripts/dev/copy_thread_exit.py)
        time.sleep(0.5)
        out[layout] = {"compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9, "compressed_bytes": int(r)}
    d = torch.empty(n, dtype=torch.uint8, device=device)
    hb = torch.from_numpy(back)
    th = tdn = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(torch.from_numpy(host)); torch.cuda.synchronize(); th = min(th, time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); hb.copy_(d); torch.cuda.synchronize(); tdn = min(tdn, time.perf_counter() - t0)
    out["pcie_GBps_measured"] = {"h2d": n / th / 1e9, "d2h": n / tdn / 1e9, "how": "torch copy of the same 1 GiB between pageable host memory and HBM, best of 3"}
    # Both directions of the link at once: the upload of the compressed bytes and the download of the output, each on its
    # own stream from its own host thread, alone and at the same time.  Measured in round 2: a stream is
    # tied to a copy engine by its first large copy -- the engine that is idle at that moment, for good -- so two streams
    # overlap only if their first copies did.  Both cases are reported: a pair bound one after the other (what a program
    # gets that never thought about it) and a pair bound together (pinned scratch: a pageable copy would not return before it
    # is done).  The calls go through ctypes: torch's copies hold Python's lock.
    try:
        import ctypes as C
        import threading
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        hip.hipStreamDestroy.argtypes = [C.c_void_p]
        nc = int(out["independent_frames"]["compressed_bytes"])
        up_dst = torch.empty(nc, dtype=torch.uint8, device=device)
        bind = 16 << 20
        p_up = torch.zeros(bind, dtype=torch.uint8).pin_memory(); p_dn = torch.zeros(bind, dtype=torch.uint8).pin_memory()

        def pair(together):
            su, sd = C.c_void_p(), C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(su), 1) == 0 and hip.hipStreamCreateWithFlags(C.byref(sd), 1) == 0
            hip.hipMemcpyAsync(up_dst.data_ptr(), p_up.data_ptr(), bind, 1, su)
            if not together:
                hip.hipStreamSynchronize(su)
            hip.hipMemcpyAsync(p_dn.data_ptr(), d.data_ptr(), bind, 2, sd)
            hip.hipStreamSynchronize(su); hip.hipStreamSynchronize(sd)
            return su, sd

        def timed(fns):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ths = [threading.Thread(target=f) for f in fns]
                [x.start() for x in ths]; [x.join() for x in ths]
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            return best
        res = {}
        for name, together in (("bound_one_after_the_other", False), ("bound_together", True)):
            su, sd = pair(together)

            def up():
                hip.hipMemcpyAsync(up_dst.data_ptr(), comp.ctypes.data, nc, 1, su); hip.hipStreamSynchronize(su)

            def down():
                hip.hipMemcpyAsync(back.ctypes.data, d.data_ptr(), n, 2, sd); hip.hipStreamSynchronize(sd)
            res[name] = (timed([up]), timed([down]), timed([up, down]))
            hip.hipStreamDestroy(su); hip.hipStreamDestroy(sd)
        t_up, t_dn, t_seq = res["bound_one_after_the_other"]
        t_both = res["bound_together"][2]
        out["pcie_GBps_measured"].update({"upload_compressed_ms": t_up * 1e3, "download_output_ms": t_dn * 1e3,
                                          "both_at_once_ms_streams_bound_one_after_the_other": t_seq * 1e3,
                                          "both_at_once_ms_streams_bound_together": t_both * 1e3,
                                          "full_duplex": bool(t_both < 0.8 * (t_up + t_dn))})
        # what a decompress (compress) call from host memory cannot beat on this box: its bytes over the link, both directions
        # at once when the copy engines allow it
        out["transfer_floor_GBps"] = n / min(t_both, t_seq) / 1e9
    except Exception as e:  # the measurement is a note next to the numbers, never a reason to lose them
        out["pcie_GBps_measured"]["duplex_error"] = repr(e)
    out["compress_GBps"] = out["one_frame"]["compress_GBps"]   # the default layout of the host entries
    out["decompress_GBps"] = out["one_frame"]["decompress_GBps"]
    log(f"end to end: {out}")
    return out


def n1_decode(codec, device, frame, src, log):
    """zstd_decompress() of ONE reference-made frame of the whole bench vector (what the
    reference's zstd_compress() writes: 8192 blocks, one window), through the large-frame pipeline,
    device-resident, checked against the plaintext."""
    import torch
    from zstdlite_amd import _lib
    n = int(src.numel())
    d_src = torch.from_numpy(np.ascontiguousarray(frame)).to(device)
    d_dst = torch.empty(n, dtype=torch.uint8, device=device)
    tbl = codec.frame_table([int(frame.size)], [n])
    before = int(_lib.lib().zl_dctx_large_frames(codec.dctx))
    best = 1e9
    host = np.ascontiguousarray(frame)
    best_h = 1e9
    for it in range(6):
        with_host = it >= 3  # the second three: the caller still holds the bytes in host memory, the block walk happens there
        d_dst.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, _ = codec.decompress(d_src, tbl, d_dst, host_copy=host if with_host else None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it % 3:
            if with_host:
                best_h = min(best_h, dt)
            else:
                best = min(best, dt)
    took = (int(_lib.lib().zl_dctx_large_frames(codec.dctx)) - before) // 2
    assert total == n and torch.equal(d_dst, src), "N1 decode mismatch"
    log(f"N1: one reference frame of {n >> 20} MiB decoded at {n / best / 1e9:.2f} GB/s (large-frame pipeline: {took == 3})")
    return {"frame_content_bytes": n, "compressed_bytes": int(frame.size), "decompress_GBps": n / best / 1e9, "ms": best * 1e3,
            "decompress_GBps_blocks_walked_on_host": n / best_h / 1e9, "ms_blocks_walked_on_host": best_h * 1e3,
            "large_frame_pipeline": took == 3, "bit_exact": True}


def c2_decode(codec, device, log, nframes=10000):
    """BASELINE config C2: zstd_decompress() of 10k independent 64 KiB frames made by the
    reference libzstd at level 3 (one third each text / iid10 / rand), decoded in one launch."""
    import torch
    from zl_testlib import Ref, have_ref
    from zstdlite_amd.device import gen_device
    if not have_ref():
        return None
    ref = Ref()
    fsz = 65536
    per = (nframes + 2) // 3
    t0 = time.perf_counter()
    srcs = {k: gen_device(k, per * fsz, device, seed=77).cpu().numpy() for k in ("text", "iid10", "rand")}
    frames, plain = [], []
    L = ref.L
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, 3)
    cap = L.ZSTD_compressBound(fsz)
    buf = C.create_string_buffer(cap)
    for i in range(nframes):
        a = srcs[("text", "iid10", "rand")[i % 3]][(i // 3) * fsz:(i // 3 + 1) * fsz]
        r = L.ZSTD_compress2(cctx, buf, cap, a.ctypes.data, fsz)
        frames.append(buf.raw[:r])
        plain.append(a)
    L.ZSTD_freeCCtx(cctx)
    log(f"C2: built {nframes} reference frames in {time.perf_counter() - t0:.1f}s")
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    want = torch.from_numpy(np.concatenate(plain)).to(device)
    d_src = torch.from_numpy(blob.copy()).to(device)
    d_dst = torch.empty(nframes * fsz, dtype=torch.uint8, device=device)
    tbl = codec.frame_table([len(f) for f in frames], [fsz] * nframes)
    best = 1e9
    for it in range(4):
        d_dst.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, _ = codec.decompress(d_src, tbl, d_dst)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
    assert total == nframes * fsz and torch.equal(d_dst, want), "C2 decode mismatch"
    km = codec.kernel_ms()
    return {"frames": nframes, "frame_bytes": fsz, "decompress_GBps": nframes * fsz / best / 1e9,
            "decode_kernel_ms": km.get("decode"), "kernel_ms": {k: km[k] for k in ("d_index", "d_entropy", "d_exec", "d_general") if k in km},
            "compressed_bytes": int(blob.size), "bit_exact": True}


class _DryCodec:
    """--dry only: stands in for the HIP codec so that the multi-rank CONTROL FLOW of a step (shard
    buffers, size all-gather, payload gather to rank 0, drains, max-over-ranks timing) can run on CPU
    ranks under gloo in the tests.  "Compression" is a copy; nothing it prints is a measurement."""

    def __init__(self):
        from zstdlite_amd.shard import CHUNK
        self.chunk = CHUNK

    def bound(self, n):
        return n + 64

    def chunk_content_sizes(self, n):
        k = max(1, -(-n // self.chunk))
        cs = np.full(k, self.chunk, dtype=np.uint64)
        cs[-1] = n - (k - 1) * self.chunk
        return cs

    def compress(self, src, dst):
        dst[:src.numel()].copy_(src)
        return dst, int(src.numel()), self.chunk_content_sizes(int(src.numel()))

    @staticmethod
    def frame_table(fs, cs):
        return None

    def decompress(self, src, table, dst):
        dst.copy_(src[:dst.numel()])
        return int(dst.numel()), None

    def kernel_ms(self):
        return {}


def _kernel_sources_hash():
    """Identity of the kernels a PMC summary was taken at (scripts/summarize_profiles.py stamps the same)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "zstdlite_amd", "csrc", "*.h*"))):  # the product kernels (csrc/experiments/ is not part of the library)
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def inproc_main(args):
    """--inproc: the multi-GPU path as the reference package could call it -- R is one process (src/init.c:90-99), so the
    sharding sits behind zl_compress2 / zl_decompressDCtx: N device slots, each uploading, coding and downloading its own
    contiguous run of 64 MiB sections from its own host thread.  One step = zstd_compress + zstd_decompress of N x --size
    bytes of HOST memory (weak scaling), PCIe both ways inside the timed region: an end-to-end number, not the
    device-resident headline.  Slot s runs on device s mod the number of visible devices."""
    import torch
    from zstdlite_amd import _lib
    from zstdlite_amd.device import gen_mixed
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    L = _lib.lib()
    ndev = torch.cuda.device_count()
    slots = args.gpus
    n = args.size * slots
    dev = torch.device("cuda", 0)
    host = np.concatenate([gen_mixed(args.size, dev)[0].cpu().numpy() for _ in range(slots)]) if slots > 1 else gen_mixed(args.size, dev)[0].cpu().numpy()
    cap = L.zl_compressBound(n)
    comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
    back = np.empty(n, dtype=np.uint8); back.fill(1)
    # several slots copy from / to these buffers at once, each over its own link: registered (pinned) with the runtime, so that
    # eight slots do not queue up behind one pageable staging path (an R caller would register its vectors the same way)
    registered = []
    if slots > 1:
        hip = C.CDLL("libamdhip64.so")
        hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        for a in (host, comp, back):
            if hip.hipHostRegister(a.ctypes.data, a.nbytes, 0) == 0:
                registered.append(a)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_setParameter(cctx, 9007, slots)
    L.zl_DCtx_setParameter(dctx, 9008, slots)
    L.zl_CCtx_setParameter(cctx, 9006, 1)  # the seek table lets the decoder deal ONE frame out over the slots
    tc = td = 0.0

    def step():
        nonlocal tc, td
        t0 = time.perf_counter()
        r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
        t1 = time.perf_counter()
        assert not _lib.is_error(r), _lib.error_name(r)
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
        t2 = time.perf_counter()
        assert r2 == n, _lib.error_name(r2)
        tc += t1 - t0
        td += t2 - t1
        return r
    for _ in range(args.warmup):
        step()
    tc = td = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        csize = step()
    elapsed = time.perf_counter() - t0
    assert np.array_equal(back, host), "round trip is not bit-exact"
    print(json.dumps({
        "metric": "GB/s zstd_compress+zstd_decompress on 1 GiB raw, level 3; ratio vs CPU", "mode": "inproc (host buffers, PCIe inside the timed region)",
        "value": n * args.steps / elapsed / 1e9, "unit": "GB/s", "n_gpus": slots, "devices_visible": ndev, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"zl_compress2 then zl_decompressDCtx of {n >> 20} MiB of host memory ({args.size >> 20} MiB per device slot), level 3, one-frame layout with seek table",
                   "parallelism": f"{slots} device slot(s) on {ndev} visible device(s) behind one call, contiguous runs of 64 MiB sections, no collective"},
        "ratio": n / csize, "compressed_bytes": int(csize), "bit_exact_roundtrip": True,
        "compress_GBps": n * args.steps / tc / 1e9, "decompress_GBps": n * args.steps / td / 1e9,
        "slots_used": {"compress": int(L.zl_cctx_last_slots(cctx)), "decompress": int(L.zl_dctx_last_slots(dctx))},
        "host_buffers_registered": len(registered)}), flush=True)
    L.zl_freeCCtx(cctx)
    L.zl_freeDCtx(dctx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=1 << 30, help="raw bytes per GPU (default 1 GiB)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-c2", action="store_true", help="skip the 10k x 64 KiB frame decode leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the one-frame, long-range, C4 and end-to-end legs")
    ap.add_argument("--dry", action="store_true", help="control flow only, CPU ranks over gloo (tests); prints no measurement")
    ap.add_argument("--inproc", action="store_true", help="ONE process: the host entries of the C-ABI spread every call over --gpus device slots (ZL_c_devices / ZL_d_devices): what an R caller gets")
    args = ap.parse_args()
    if args.inproc:
        return inproc_main(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched without torchrun: start the N ranks ourselves.  This parent has not touched the GPU
        # (no torch import yet), so it only waits for the child launcher and passes its exit code on.
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from zstdlite_amd.device import gen_mixed
    from zstdlite_amd.shard import gather_compressed

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.dry:
        device = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
        if torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible device(s)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=device)
        sync = torch.cuda.synchronize
    ids = None
    if world > 1:
        # every rank on its own device, and the communicator as large as the flag says
        ids = [None] * world
        dist.all_gather_object(ids, (local_rank, "cpu" if args.dry else str(torch.cuda.get_device_properties(local_rank).uuid)
                                     if hasattr(torch.cuda.get_device_properties(local_rank), "uuid") else local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        assert len({i[0] for i in ids}) == world, f"ranks share a device: {ids}"
        if not args.dry:
            assert len({i[1] for i in ids}) == world, f"ranks share a device: {ids}"

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    n = args.size
    # ONE vector of world * n bytes, rank r holds bytes plan_shards(world * n, world)[r] of it (shard r has its own generator
    # seeds; shard 0 is the one-GPU vector): the ranks do not compress copies of the same gigabyte
    from zstdlite_amd.shard import plan_shards
    shards = [(r * n, (r + 1) * n) for r in range(world)]
    if n % (128 << 10) == 0:
        assert shards == plan_shards(world * n, world), "the ranks' byte ranges are the shard plan of the whole vector"
    src, kinds, quarter = gen_mixed(n, device, shard=rank)
    if args.dry:
        codec = _DryCodec()
    else:
        from zstdlite_amd.device import DeviceCodec
        codec = DeviceCodec(level=3, profile=True)
    # two compressed buffers (and two gather buffers on rank 0): the gather of one step's shard runs
    # while the next step computes; every gather is waited for inside the timed region
    comps = [torch.empty(codec.bound(n), dtype=torch.uint8, device=device) for _ in range(2 if world > 1 else 1)]
    comp = comps[0]
    back = torch.empty(n, dtype=torch.uint8, device=device)
    content = codec.chunk_content_sizes(n)
    gathered = [None, None]
    pending = [[], []]
    step_no = 0
    kms_sum = {}

    def drain(slot):
        if pending[slot]:
            for w in pending[slot]:
                w.wait()  # (RCCL: orders torch's current stream behind the transfer, does not block the host)
            # the codec launches on its own streams, so the host itself must see the transfer finished
            # before the next compression may overwrite this buffer
            if not args.dry:
                torch.cuda.current_stream().synchronize()
        pending[slot] = []

    def step():
        nonlocal step_no
        slot = step_no % len(comps)
        step_no += 1
        drain(slot)  # this buffer's previous shard has left (no-op on one GPU)
        buf = comps[slot]
        _, total, fsizes = codec.compress(src, buf)
        if world > 1:
            # the path's one exchange step: compressed shard sizes, then payloads to rank 0 over xGMI
            gathered[slot], _, pending[slot] = gather_compressed(buf, total, root=0, out=gathered[slot], wait=False)
        tbl = codec.frame_table(fsizes, content)
        out_total, _ = codec.decompress(buf, tbl, back)
        assert out_total == n
        return total

    for _ in range(args.warmup):
        step()
    drain(0)
    drain(1)
    kms_sum = {}
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        csize = step()
        for k, v in codec.kernel_ms().items():
            kms_sum[k] = kms_sum.get(k, 0.0) + v
    drain(0)
    drain(1)
    sync()
    own = time.perf_counter() - t0  # this rank's own time, before waiting for the others
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_gbps = [n * args.steps / own / 1e9]
    gather_ms = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([per_rank_gbps[0]], dtype=torch.float64, device=device)
        allr = torch.zeros(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(allr, mine)
        per_rank_gbps = [float(x) for x in allr.cpu().tolist()]
        # the exchange step on its own (not overlapped): sizes + payloads of one step to rank 0
        dist.barrier()
        sync()
        tg = time.perf_counter()
        gathered[0], _ = gather_compressed(comps[0], csize, root=0, out=gathered[0])
        sync()
        dist.barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
    assert torch.equal(back, src), "round trip is not bit-exact"
    # ---- each direction on its own, same bracket (barrier + synchronize, max over ranks): north_star's target is the
    # AGGREGATE zstd_decompress() rate; compression alone next to it
    def timed(fn):
        for _ in range(min(2, args.warmup)):
            fn()
        if world > 1:
            dist.barrier()
        sync()
        t0d = time.perf_counter()
        for _ in range(args.steps):
            fn()
        sync()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0d
        if world > 1:
            tm = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = float(tm.item())
        return el / args.steps
    _, total_d, fsizes_d = codec.compress(src, comp)
    tbl_d = codec.frame_table(fsizes_d, content)
    t_dec = timed(lambda: codec.decompress(comp, tbl_d, back))
    assert torch.equal(back, src), "decompression alone is not bit-exact"
    t_cmp = timed(lambda: codec.compress(src, comp))
    ms_per_step = elapsed / args.steps * 1e3
    kms = {k: v / args.steps for k, v in kms_sum.items()}

    if rank == 0 and args.dry:
        print(json.dumps({"dry": True, "metric": "control flow only (no measurement)", "value": None, "n_gpus": world,
                          "world_size": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
                          "bytes_per_rank": n, "per_rank_ok": len(per_rank_gbps) == world, "vector_bytes": world * n, "shard_of_rank": [list(x) for x in shards]}), flush=True)
    elif rank == 0:
        value = world * n / (ms_per_step / 1e3) / 1e9
        t_comp = sum(kms.get(k, 0.0) for k in ("match", "entropy", "scan", "gather"))
        out = {
            "metric": "GB/s zstd_compress+zstd_decompress on 1 GiB raw, level 3; ratio vs CPU",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"zstd_compress() then zstd_decompress() of a {n >> 20} MiB raw vector per GPU, level 3 (BASELINE configs[2] + its decode); vector = 4 equal parts rand/iid10/text/int20 (SURVEY 8d); one independent frame per 128 KiB chunk (frame content 131072 bytes)" + (f"; the {world} ranks hold the {world} shards of ONE {world * n >> 20} MiB vector (own seeds per shard)" if world > 1 else ""),
                       "bytes_per_gpu": n, "chunk_bytes": 128 << 10, "frame_content_bytes": 128 << 10, "frames_per_gpu": int(len(content)), "vector_bytes": world * n, "shard_of_rank": [list(x) for x in shards],
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective" + ("; compressed shards gathered to rank 0 over RCCL" if world > 1 else "")},
            "ratio": n / csize, "compressed_bytes": int(csize), "bit_exact_roundtrip": True,
            "decompress_only_GBps_aggregate": world * n / t_dec / 1e9, "compress_only_GBps_aggregate": world * n / t_cmp / 1e9,
            "compress_GBps_kernels": (n / (t_comp / 1e3) / 1e9) if t_comp else None,
            "decompress_GBps_kernel": (n / (kms["decode"] / 1e3) / 1e9) if kms.get("decode") else None,
            "kernel_ms": kms,
            "per_rank_GBps": per_rank_gbps, "gather_ms_standalone": gather_ms,
            # what the communicator itself says (SCALE records can be checked against it): backend "nccl" is RCCL on ROCm
            "distributed": {"backend": dist.get_backend() if world > 1 else None, "world_size_seen_by_backend": dist.get_world_size() if world > 1 else 1,
                            "rank_devices": [{"rank": r, "local_device": i[0], "uuid": str(i[1])} for r, i in enumerate(ids)] if ids else
                            [{"rank": 0, "local_device": local_rank, "uuid": str(getattr(torch.cuda.get_device_properties(local_rank), "uuid", local_rank))}]},
        }
        # ---- roofline of the dominant kernel (HIP events on the launch stream, averaged over the timed steps)
        if kms:
            single = {k: v for k, v in kms.items() if k != "decode"}  # "decode" is the sum of its four kernels
            dom = max(single, key=single.get)
            alg = n + csize  # N_in + N_comp (compress) == N_out + N_comp (decompress)
            ach = alg / (kms[dom] / 1e3) / 1e9
            # HBM bytes per launch of that kernel from the PMC passes kept under profiles/ (separate rocprofv3 --pmc runs of
            # this same command, scripts/profile_round.sh).  The summary names the kernel sources it was taken at: when
            # they have changed since, the number describes another kernel and is not reported.
            traffic, traffic_from = None, None
            pmc = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1:] or [""]
            pmc = pmc[0]
            if os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    if pj.get("_kernel_sources_sha256") == _kernel_sources_hash():
                        traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
                        traffic_from = os.path.basename(pmc)
                except Exception:
                    traffic = None
            # the practical ceiling next to the 8 TB/s specification: a device-to-device copy of the same vector
            cp = torch.empty_like(src)
            tcp = 1e9
            for _ in range(5):
                torch.cuda.synchronize(); t0c = time.perf_counter(); cp.copy_(src); torch.cuda.synchronize(); tcp = min(tcp, time.perf_counter() - t0c)
            del cp
            out["roofline"] = {"bound": "hbm", "kernel": {"match": "zl_match_auto_kernel<false, false, false, false, false>", "entropy": "zl_entropy_kernel", "scan": "zl_scan_sizes_kernel",
                                                           "gather": "zl_gather_kernel", "d_index": "zl2_index_kernel", "d_entropy": "zl2_seq_kernel",
                                                           "d_exec": "zl2_exec_kernel<false, false>", "d_general": "zl_decode_kernel"}[dom],
                               "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                               "traffic_source": traffic_from, "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": kms[dom],
                               "device_copy_ceiling_GBps": 2 * n / tcp / 1e9, "frac_of_copy_ceiling": ach / (2 * n / tcp / 1e9)}
        # ---- per-input ratios next to BASELINE.md section 2
        per = {}
        for i, k in enumerate(kinds):
            part = src[i * quarter:(i + 1) * quarter if i < 3 else n]
            _, tot, _ = codec.compress(part.contiguous(), comp)
            per[k] = {"bytes": int(part.numel()), "compressed": int(tot), "ratio": part.numel() / tot}
        out["per_input"] = per
        if world == 1 and not args.no_extra:
            for name, leg in (("one_frame_layout", lambda: one_frame_leg(device, src, log)), ("per_input_full", lambda: per_input_full_leg(codec, device, n, log)),
                              ("long_range", lambda: long_range_leg(device, log)), ("levels", lambda: levels_leg(device, src, log)),
                              ("c4_small_objects_dict", lambda: c4_leg(log)), ("single_object_host_to_host", lambda: single_object_leg(log))):
                try:
                    out[name] = leg()
                except Exception as e:  # informative legs; never lose the headline line to them
                    out[name] = {"error": repr(e)}
            # the DEFAULT layout of the host entries (what an R caller gets) has its own roofline entry next to the headline's
            if isinstance(out.get("one_frame_layout"), dict) and "roofline" in out["one_frame_layout"]:
                out["roofline_one_frame"] = out["one_frame_layout"].pop("roofline")
                # the same round trip in the DEFAULT layout of zl_compress2 (one frame): what an R caller's bytes look like
                of = out["one_frame_layout"]
                out["value_one_frame"] = 1.0 / (1.0 / of["compress_GBps"] + 1.0 / of["decompress_GBps"])
        if world == 1 and not args.no_cpu:
            # ---- cpu_baseline leg: the only place bench.py touches oracle/ (the reference libzstd
            # built into oracle/_ref): timed on the host cores, and -- for BASELINE config C2 -- used
            # to produce the 10k x 64 KiB level-3 frames whose GPU decode is checked against plaintext
            host = src.cpu().numpy()
            if not args.no_extra:
                try:
                    out["end_to_end"] = end_to_end_leg(host, device, log)
                except Exception as e:
                    out["end_to_end"] = {"error": repr(e)}
                try:  # the streaming entries on the same vector (scripts/stream_bench.py)
                    sys.path.insert(0, os.path.join(ROOT, "scripts"))
                    import stream_bench
                    out["streaming"] = stream_bench.main(log=log, host=host)
                except Exception as e:
                    out["streaming"] = {"error": repr(e)}
            _, tot_f, fs_f = codec.compress(src, comp)  # the per-chunk frames of the vector: input of the threaded CPU decode
            cb = cpu_baseline(host, log, frames=(comp[:tot_f].cpu().numpy(), np.asarray(fs_f, dtype=np.int64), np.asarray(content, dtype=np.int64)))
            if cb is not None:
                ref_frame = cb.pop("_frame")
                out["cpu_baseline"] = cb
                out["ratio_vs_cpu"] = {"speedup_roundtrip": value / cb["value"], "compressed_size_ours_over_cpu": csize / cb["compressed_bytes"]}
                if not args.no_c2:
                    try:
                        out["c2_frames64k"] = c2_decode(codec, device, log)
                    except Exception as e:  # informative leg; never lose the headline line to it
                        out["c2_frames64k"] = {"error": repr(e)}
                try:
                    out["n1_large_frame"] = n1_decode(codec, device, ref_frame, src, log)
                    out["n1_large_frame"]["cpu_decompress_GBps"] = cb["decompress_GBps"]
                except Exception as e:
                    out["n1_large_frame"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
