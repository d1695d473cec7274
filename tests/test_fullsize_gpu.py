"""BASELINE.json's full sizes, checked through size-independent properties (the oracle would take
minutes here): 1 GiB round trip bit-exact and deterministic on the device-resident entry points,
every frame regenerating exactly its chunk, corruption of any one frame reported for that frame
only, and config C2 (10k independent 64 KiB reference-made frames) decoded in one launch."""
import ctypes as C

import numpy as np
import pytest
import torch

from zstdlite_amd.device import CHUNK, DeviceCodec, gen_device, gen_mixed
from zl_testlib import have_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    c = DeviceCodec()
    yield c
    c.close()


def test_one_gib_roundtrip_deterministic(codec):
    dev = torch.device("cuda", 0)
    n = 1 << 30
    src, kinds, q = gen_mixed(n, dev)
    comp, total, fs = codec.compress(src)
    assert total == int(fs.sum()) and len(fs) == n // CHUNK
    assert total <= codec.bound(n)
    digest1 = int(comp[:total].to(torch.int64).sum().item())
    back = torch.empty_like(src)
    tbl = codec.frame_table(fs, codec.chunk_content_sizes(n))
    got, res = codec.decompress(comp, tbl, back)
    assert got == n and (res == CHUNK).all()
    assert torch.equal(back, src)
    # determinism (tests/testthat/test-cctx.R): same bytes on a second run with a fresh context
    c2 = DeviceCodec()
    comp2, total2, fs2 = c2.compress(src)
    assert total2 == total and (fs2 == fs).all() and torch.equal(comp2[:total], comp[:total])
    assert int(comp2[:total].to(torch.int64).sum().item()) == digest1
    c2.close()
    # ratio sanity against BASELINE.md section 2 (reference level 3 on 256 MiB of each input)
    ref_bytes = {"rand": 268441610, "iid10": 131457759, "text": 63726261, "int20": 210971766}
    starts = np.concatenate([[0], np.cumsum(fs)])
    per = n // 4 // CHUNK
    for i, k in enumerate(kinds):
        ours = int(starts[(i + 1) * per] - starts[i * per])
        assert ours <= ref_bytes[k] * 1.05, (k, ours, ref_bytes[k])
    # damage one byte in the middle of three frames.  Without a content checksum a flipped bit inside a
    # raw block -- or inside a Huffman stream that happens to resynchronise -- is not detectable, so
    # this part runs with include_checksum=TRUE: exactly the damaged frames must fail, every other
    # frame must still be bit-exact (tests/testthat/test-checksums.R at scale).
    c3 = DeviceCodec(include_checksum=True)
    m = 64 << 20
    part = torch.cat([src[i * q: i * q + m // 4] for i in range(4)])
    compc, totalc, fsc = c3.compress(part)
    assert totalc == int(fsc.sum())
    tblc = c3.frame_table(fsc, c3.chunk_content_sizes(m))
    startsc = np.concatenate([[0], np.cumsum(fsc)])
    victims = [3, 200, 300, 511]  # one frame of each input kind
    for v in victims:
        compc[int(startsc[v] + fsc[v] // 2)] ^= 0x40
    backc = torch.zeros(m, dtype=torch.uint8, device=dev)
    from zstdlite_amd._lib import FrameT
    L = c3.L
    r = L.zl_decompress_device_async(c3.dctx, C.c_void_p(backc.data_ptr()), C.c_void_p(compc.data_ptr()),
                                     tblc.ctypes.data_as(C.POINTER(FrameT)), len(tblc), None)
    assert r == 0
    resc = (C.c_size_t * len(tblc))()
    first = L.zl_decompress_device_finish(c3.dctx, resc)
    assert first > (1 << 63)  # an error code, not a size
    resc = np.frombuffer(resc, dtype=np.uint64)
    failed = set(np.nonzero(resc > (1 << 63))[0].tolist())
    assert failed == set(victims)
    ok = np.ones(len(tblc), dtype=bool)
    ok[victims] = False
    keep = torch.from_numpy(np.repeat(ok, CHUNK)).to(dev)
    assert torch.equal(backc[keep], part[keep])
    c3.close()


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (reference libzstd) to make the frames")
def test_c2_ten_thousand_64k_reference_frames(codec):
    from zl_testlib import Ref
    ref = Ref()
    dev = torch.device("cuda", 0)
    nframes, fsz = 10000, 65536
    per = (nframes + 2) // 3
    srcs = {k: gen_device(k, per * fsz, dev, seed=5).cpu().numpy() for k in ("text", "iid10", "rand")}
    L = ref.L
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, 3)
    cap = L.ZSTD_compressBound(fsz)
    buf = C.create_string_buffer(cap)
    frames, plain = [], []
    for i in range(nframes):
        a = srcs[("text", "iid10", "rand")[i % 3]][(i // 3) * fsz:(i // 3 + 1) * fsz]
        r = L.ZSTD_compress2(cctx, buf, cap, a.ctypes.data, fsz)
        frames.append(buf.raw[:r])
        plain.append(a)
    L.ZSTD_freeCCtx(cctx)
    blob = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    want = torch.from_numpy(np.concatenate(plain)).to(dev)
    out = torch.empty(nframes * fsz, dtype=torch.uint8, device=dev)
    tbl = codec.frame_table([len(f) for f in frames], [fsz] * nframes)
    total, res = codec.decompress(blob, tbl, out)
    assert total == nframes * fsz and (res == fsz).all()
    assert torch.equal(out, want)


def test_one_gib_ten_times_on_fresh_contexts():
    """Stress (a 1-in-15 wrong byte was found by hand in round 1): the 1 GiB round trip ten times over, a
    fresh pair of contexts each time, both layouts alternating -- same compressed bytes every time, bit-exact output."""
    dev = torch.device("cuda", 0)
    n = 1 << 30
    src, _, _ = gen_mixed(n, dev)
    back = torch.empty_like(src)
    digests = {}
    for it in range(10):
        layout = ("frames", "one_frame")[it & 1]
        c = DeviceCodec(frame_layout=layout)
        comp, total, fs = c.compress(src)
        dig = (total, int(comp[: total // 8 * 8].view(torch.int64).sum().item()), int(comp[total // 8 * 8: total].to(torch.int64).sum().item()))
        assert digests.setdefault(layout, dig) == dig, (it, layout)
        back.zero_()
        if layout == "frames":
            got, _ = c.decompress(comp, c.frame_table(fs, c.chunk_content_sizes(n)), back)
        else:
            got, _ = c.decompress(comp, c.frame_table([total], [n]), back, chunk_sizes=fs)
        assert got == n and torch.equal(back, src), (it, layout)
        c.close()
        del comp


def test_host_entry_sections_write_the_same_bytes_as_one_launch():
    """zl_compress2 of a large input uploads, compresses and downloads in sections on several streams (ZL_c_pipeline);
    what it writes must not depend on that: same bytes as the one-stream path and as the device entry, both layouts."""
    from zstdlite_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = 320 << 20
    src, _, _ = gen_mixed(n, dev)
    host = src.cpu().numpy()
    cap = L.zl_compressBound(n)
    outs = {}
    for layout, code in (("one_frame", 2), ("frames", 1)):
        for pipe in (1, 0):
            cctx = L.zl_createCCtx()
            L.zl_CCtx_setParameter(cctx, 9003, code)
            L.zl_CCtx_setParameter(cctx, 9004, pipe)
            buf = np.empty(cap, dtype=np.uint8)
            r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n)
            assert not _lib.is_error(r), _lib.error_name(r)
            outs[(layout, pipe)] = buf[:r].copy()
            L.zl_freeCCtx(cctx)
        assert np.array_equal(outs[(layout, 1)], outs[(layout, 0)]), layout
        c = DeviceCodec(frame_layout=layout)
        comp, total, _ = c.compress(src)
        assert total == len(outs[(layout, 1)]) and np.array_equal(comp[:total].cpu().numpy(), outs[(layout, 1)]), layout
        c.close()
    dctx = L.zl_createDCtx()
    back = np.empty(n, dtype=np.uint8)
    for key, blob in outs.items():
        back.fill(0)
        assert L.zl_decompressDCtx(dctx, back.ctypes.data, n, blob.ctypes.data, len(blob)) == n and np.array_equal(back, host), key
    L.zl_freeDCtx(dctx)


def _host_roundtrip(L, host, layout, devices, seek):
    from zstdlite_amd import _lib
    n = host.size
    cap = L.zl_compressBound(n)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_setParameter(cctx, 9003, layout)
    L.zl_CCtx_setParameter(cctx, 9006, seek)
    assert L.zl_CCtx_setParameter(cctx, 9007, devices) == 0 and L.zl_DCtx_setParameter(dctx, 9008, devices) == 0
    buf = np.empty(cap, dtype=np.uint8)
    r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n)
    assert not _lib.is_error(r), _lib.error_name(r)
    back = np.zeros(n, dtype=np.uint8)
    r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, buf.ctypes.data, r)
    assert r2 == n, _lib.error_name(r2)
    assert np.array_equal(back, host)
    used = (int(L.zl_cctx_last_slots(cctx)), int(L.zl_dctx_last_slots(dctx)))
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
    return buf[:r].copy(), used


def test_device_slots_behind_the_host_entries():
    """SURVEY 8e inside one process (the reference is one process: src/init.c:90-99): ZL_c_devices / ZL_d_devices deal the
    sections of a large host buffer out over device slots.  The bytes written must not depend on the number of slots; a
    slot count above the number of devices puts several slots on one device, which is how one GPU tests the path (with
    more than one device the same test spreads over them)."""
    from zstdlite_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = (640 << 20) + 12345
    host = gen_mixed(n, dev)[0].cpu().numpy()
    for layout in (2, 1):           # one frame, independent frames
        for seek in (1, 0):         # (without a seek table one frame is decoded by one device)
            one, used = _host_roundtrip(L, host, layout, 1, seek)
            # one device: compression pipelines inside its one slot; decompression overlaps upload, decode and download by
            # running three slots on the device (ZL_d_pipeline) -- unless it is ONE frame without a seek table
            assert used == (1, 1 if (layout == 2 and not seek) else 3), (layout, seek, used)
            for slots in (2, 3):
                got, used = _host_roundtrip(L, host, layout, slots, seek)
                assert np.array_equal(got, one), (layout, seek, slots)
                assert used[0] == slots and (used[1] == 1 if (layout == 2 and not seek) else used[1] >= slots), (layout, seek, slots, used)
    # BASELINE config C5's shape: eight slots (on one device here) write the bytes one slot writes, for an input of 8 x 128 MiB + a tail
    got8, used8 = _host_roundtrip(L, host, 2, 8, 1)
    assert np.array_equal(got8, _host_roundtrip(L, host, 2, 1, 1)[0]) and used8[0] == 8


def test_deep_levels_through_the_pipelined_host_entry():
    """Levels >= 6 (csrc/zl_deep.hip) through the host entry's pipeline: sections of 64 MiB on three streams with a scratch set each (the
    deep finder sorts a section at a time), and dealt out over device slots: the bytes must be those of the one-stream device entry, whatever
    the staging, and the reference must read them -- one frame and independent frames."""
    import zstdlite_amd as z
    from zstdlite_amd import _lib
    from zl_testlib import Ref
    if not have_ref():
        pytest.skip("needs the reference build (oracle/_ref)")
    ref = Ref()
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = (300 << 20) + 4321
    src = gen_mixed(n, dev)[0]
    host = src.cpu().numpy()
    for level, layout in ((9, 2), (6, 1), (17, 2)):  # (17: the price-based parse of csrc/zl_opt.hip over two search passes, round 6)
        outs = []
        for slots in (1, 3):
            cctx = L.zl_createCCtx()
            L.zl_CCtx_setParameter(cctx, 100, level); L.zl_CCtx_setParameter(cctx, 9003, layout); L.zl_CCtx_setParameter(cctx, 9007, slots)
            cap = L.zl_compressBound(n)
            buf = np.empty(cap, dtype=np.uint8)
            r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n)
            assert not _lib.is_error(r), _lib.error_name(r)
            outs.append(buf[:r].copy())
            L.zl_freeCCtx(cctx)
        assert np.array_equal(outs[0], outs[1]), (level, layout)
        c = DeviceCodec(level=level, frame_layout={1: "frames", 2: "one_frame"}[layout])
        comp, total, _ = c.compress(src)
        assert total == len(outs[0]) and np.array_equal(comp[:total].cpu().numpy(), outs[0]), (level, layout)
        c.close()
        blob = outs[0].tobytes()
        if layout == 2:
            assert ref.decompress(blob, n) == host.tobytes()
        assert z.zstd_decompress(blob) == host.tobytes()


def test_device_slots_on_every_visible_device():
    """The multi-device leg of the test above (ZL_c_devices = 0: every visible device): needs more than one GPU; the pool's boxes
    have one, so it skips there -- on its own, so that the one-device body above reports what it is: passed."""
    from zstdlite_amd import _lib
    if torch.cuda.device_count() < 2:
        pytest.skip("one device: the slots shared it; the multi-device leg needs more")
    L = _lib.lib()
    n = (640 << 20) + 12345
    host = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy()
    one = _host_roundtrip(L, host, 2, 1, 1)[0]
    assert np.array_equal(_host_roundtrip(L, host, 2, 0, 1)[0], one)  # 0: every visible device


def test_checksum_and_dictionary_contexts_write_one_frame_at_320_mib():
    """The host entry's pipelined path (inputs above 192 MiB go up, through the kernels and down in overlapping 64 MiB
    sections): with a checksum or a dictionary it writes ONE frame as well -- the XXH64 of the whole content is computed on a
    host thread beside the upload (zstd.c:27733), the dictionary is history in front of the first chunk (zstd.c:27550).
    Checked with the reference's own first-frame helper and decoder."""
    import zstdlite_amd as z
    from zl_testlib import Ref
    if not have_ref():
        pytest.skip("needs the reference build (oracle/_ref)")
    ref = Ref()
    n = 320 << 20
    data = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
    comp = z.zstd_compress(data, include_checksum=True)
    assert ref.L.ZSTD_findFrameCompressedSize(comp, len(comp)) == len(comp) and ref.L.ZSTD_getFrameContentSize(comp, len(comp)) == n
    assert comp[:-4] == z.zstd_compress(data)[:4] + bytes([z.zstd_compress(data)[4] | 4]) + z.zstd_compress(data)[5:]
    assert ref.decompress(comp, n) == data
    assert z.zstd_decompress(comp) == data
    bad = bytearray(comp)
    bad[-3] ^= 1
    with pytest.raises(z.ZstdError, match="doesn't match checksum"):
        z.zstd_decompress(bytes(bad))
    dic = data[(64 << 20) + 1000:(64 << 20) + 60000]  # raw content that the first chunk does not hold by itself
    cd = z.zstd_compress(data, dict=dic)
    assert ref.L.ZSTD_findFrameCompressedSize(cd, len(cd)) == len(cd)
    assert ref.decompress(cd, n, dict=dic) == data
    assert z.zstd_decompress(cd, dict=dic) == data


def test_c4_hundred_thousand_objects_with_a_dictionary():
    """BASELINE config C4 at its full size: 100 000 small objects, one trained dictionary, through the batch entries of the
    C-ABI on arrays packed once.  Bit-exact round trip, one frame in fifty through the reference decoder with the dictionary,
    compressed size within 3 % of the reference's for the same objects and dictionary."""
    import ctypes as C
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from zstdlite_amd import _lib
    from zl_testlib import Ref
    if not have_ref():
        pytest.skip("needs the reference build (oracle/_ref)")
    ref = Ref()
    L = _lib.lib()
    items = bench.c4_items(100000)
    d = ref.train_dict(items[:2000], 5000)
    B = bench.C4Batch(items, L.zl_compressBound)
    vp, sp = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    L.zl_CCtx_loadDictionary(cctx, d, len(d)); L.zl_DCtx_loadDictionary(dctx, d, len(d))
    assert L.zl_compress_batch(cctx, B.n, B.p_comp.ctypes.data_as(vp), B.caps.ctypes.data_as(sp), B.p_src.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.csize.ctypes.data_as(sp)) == 0
    assert not (B.csize > B.caps).any()
    assert L.zl_decompress_batch(dctx, B.n, B.p_back.ctypes.data_as(vp), B.sizes.ctypes.data_as(sp), B.p_comp.ctypes.data_as(vp), B.csize.ctypes.data_as(sp), B.bsize.ctypes.data_as(sp)) == 0
    assert (B.bsize == B.sizes).all() and np.array_equal(B.back, B.blob)
    theirs = 0
    for i in range(0, B.n, 50):
        f = B.comp[int(B.coff[i]):int(B.coff[i]) + int(B.csize[i])].tobytes()
        assert ref.decompress(f, len(items[i]), dict=d) == items[i]
        theirs += len(ref.compress(items[i], level=3, dict=d))
    assert int(B.csize[::50].sum()) <= theirs * 1.03
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)


def test_one_frame_with_a_checksum_is_still_decoded_on_every_slot():
    """One frame + content checksum + seek table, decoded on two device slots: the first slot's run starts with the frame's
    header -- which says "checksum" -- but holds no footer; only the last run ends in the 4 bytes.  Until round 3 the first
    run was walked as if its last chunk were 4 bytes shorter: the walk failed, the call fell back to ONE slot and the header
    walk (same bytes, a third of the speed, and nothing said so).  Both slots must be used, the checksum must still be
    verified over the whole: a flipped checksum byte is "doesn't match checksum", a flipped content byte is an error."""
    from zstdlite_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = (200 << 20) + 4321
    host = gen_mixed(n, dev)[0].cpu().numpy()
    cap = L.zl_compressBound(n)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    try:
        L.zl_CCtx_setParameter(cctx, 9003, 2); L.zl_CCtx_setParameter(cctx, 9006, 1); L.zl_CCtx_setParameter(cctx, 201, 1)
        assert L.zl_DCtx_setParameter(dctx, 9008, 2) == 0
        buf = np.empty(cap, dtype=np.uint8)
        r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n)
        assert not _lib.is_error(r), _lib.error_name(r)
        back = np.zeros(n, dtype=np.uint8)
        assert L.zl_decompressDCtx(dctx, back.ctypes.data, n, buf.ctypes.data, r) == n
        assert np.array_equal(back, host)
        assert int(L.zl_dctx_last_slots(dctx)) == 2
        first = int(L.zl_findFrameCompressedSize(buf.ctypes.data, r))  # the data frame; the seek table is a skippable frame behind it
        assert 0 < first < r
        bad = buf[:r].copy(); bad[first - 2] ^= 0x10
        e = L.zl_decompressDCtx(dctx, back.ctypes.data, n, bad.ctypes.data, r)
        assert _lib.is_error(e) and "match checksum" in _lib.error_name(e), _lib.error_name(e)
        bad = buf[:r].copy(); bad[first // 2] ^= 0x01
        assert _lib.is_error(L.zl_decompressDCtx(dctx, back.ctypes.data, n, bad.ctypes.data, r))
        # ... and without the table: the frame is walked, its bytes come down while the last stage runs and a second thread hashes
        # them as they land (zl3_hint::sink_hash), the rest behind the call
        L.zl_CCtx_setParameter(cctx, 9006, 0)
        r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n)
        assert not _lib.is_error(r) and int(L.zl_findFrameCompressedSize(buf.ctypes.data, r)) == r
        back.fill(0)
        assert L.zl_decompressDCtx(dctx, back.ctypes.data, n, buf.ctypes.data, r) == n and np.array_equal(back, host)
        bad = buf[:r].copy(); bad[r - 1] ^= 0x80
        e = L.zl_decompressDCtx(dctx, back.ctypes.data, n, bad.ctypes.data, r)
        assert _lib.is_error(e) and "match checksum" in _lib.error_name(e), _lib.error_name(e)
        assert L.zl_decompressDCtx(dctx, back.ctypes.data, n, buf.ctypes.data, r) == n and np.array_equal(back, host)
    finally:
        L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)


def test_device_entry_decodes_one_frame_of_more_than_4_gib_in_runs():
    """The device-resident entries on ONE frame of 4 GiB and more: the large-frame pipeline plans below 4 GiB and the general
    kernel is one workgroup per frame -- 400 s for this input in round 2.  With the chunk hint (zl_decompress_device_hint) the
    frame is decoded in runs of 16 sections (decode_hinted_frame_in_runs, zl_api.hip): seconds.  Bit-exact; with a content
    checksum, which covers the whole and is checked behind the runs: a flipped footer byte must be reported."""
    import time
    from zstdlite_amd.device import DeviceCodec
    dev = torch.device("cuda", 0)
    n = (4 << 30) + (300 << 20) + 777
    piece = gen_mixed(1 << 30, dev)[0]
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    for off in range(0, n, piece.numel()):   # (sections are self-contained: repeating the GiB does not make far matches)
        k = min(piece.numel(), n - off)
        src[off:off + k] = piece[:k]
        src[off:off + 4096] ^= (off >> 30) + 1
    del piece
    codec = DeviceCodec(frame_layout="one_frame", include_checksum=True)
    try:
        comp, total, fs = codec.compress(src)
        out = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        got, _ = codec.decompress(comp, codec.frame_table([total], [n]), out, chunk_sizes=fs)
        dt = time.time() - t0
        assert got == n and torch.equal(out, src)
        assert dt < 30, f"{dt:.1f} s: the frame went to the general kernel"
        bad = comp[:total].clone()
        bad[total - 2] ^= 0x40
        with pytest.raises(Exception, match="match checksum"):
            codec.decompress(bad, codec.frame_table([total], [n]), out, chunk_sizes=fs)
        bad[total - 2] ^= 0x40
        bad[total // 2] ^= 0x01  # damage inside a chunk: reported by its run, at once
        t0 = time.time()
        with pytest.raises(Exception):
            codec.decompress(bad, codec.frame_table([total], [n]), out, chunk_sizes=fs)
        assert time.time() - t0 < 30
        wrong = fs.copy(); wrong[5] += 1; wrong[6] -= 1   # a hint that does not describe the frame is ignored, not believed
        small = DeviceCodec(frame_layout="one_frame")
        c2, t2, f2 = small.compress(src[: 3 << 20])
        o2 = torch.zeros(3 << 20, dtype=torch.uint8, device=dev)
        w2 = f2.copy(); w2[5] += 1; w2[6] -= 1
        assert small.decompress(c2, small.frame_table([t2], [3 << 20]), o2, chunk_sizes=w2)[0] == 3 << 20 and torch.equal(o2, src[: 3 << 20])
        small.close()
    finally:
        codec.close()


def test_reference_frame_of_more_than_4_gib_is_decoded_in_runs():
    """What the reference's zstd_serialize writes for an R object of 4 GiB and more: ONE frame, no seek table, no chunk hint,
    matches as far back as the window.  The large-frame pipeline plans below 4 GiB and the general kernel is one workgroup --
    11 MB/s, 400 s for this frame.  zl_decompressDCtx decodes it in runs of 8192 blocks (zl_api.hip
    decode_walked_frame_in_runs): the runs must have been taken, the bytes must be the input, with a content checksum (checked
    over the whole); a flipped byte in the middle is an error, as it is for the reference."""
    import time
    from zstdlite_amd import _lib
    from zl_testlib import Ref
    if not have_ref():
        pytest.skip("needs the reference build (oracle/_ref)")
    ref = Ref()
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = (4 << 30) + (200 << 20) + 4321
    piece = gen_mixed(1 << 30, dev)[0].cpu().numpy()
    host = np.empty(n, dtype=np.uint8)
    for off in range(0, n, piece.size):   # (the window of level 3 is 2 MiB: repeating the GiB makes no matches)
        k = min(piece.size, n - off)
        host[off:off + k] = piece[:k]
        host[off:off + 4096] ^= (off >> 30) + 1
    del piece
    cap = ref.L.ZSTD_compressBound(n)
    comp = np.empty(cap, dtype=np.uint8)
    cctx = ref.L.ZSTD_createCCtx()
    ref.L.ZSTD_CCtx_setParameter(cctx, ref.ZSTD_c_compressionLevel, 3); ref.L.ZSTD_CCtx_setParameter(cctx, ref.ZSTD_c_checksumFlag, 1)
    r = ref.L.ZSTD_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n)
    ref.L.ZSTD_freeCCtx(cctx)
    assert not ref.L.ZSTD_isError(r)
    dctx = L.zl_createDCtx()
    try:
        back = np.zeros(n, dtype=np.uint8)
        t0 = time.time()
        got = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
        dt = time.time() - t0
        assert got == n, _lib.error_name(got)
        assert int(L.zl_dctx_frame_runs(dctx)) >= 5
        assert np.array_equal(back, host)
        assert dt < 60, f"{dt:.1f} s"
        # the device-resident entry, with nothing but device pointers: the compressed bytes come down once for the run plan
        from zstdlite_amd.device import DeviceCodec
        codec = DeviceCodec()
        d_src = torch.from_numpy(comp[:r]).to(dev)
        d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
        t0 = time.time()
        got, _ = codec.decompress(d_src, codec.frame_table([r], [n]), d_out)
        assert got == n and time.time() - t0 < 60
        assert int(L.zl_dctx_frame_runs(codec.dctx)) >= 5
        back[:] = d_out.cpu().numpy()
        assert np.array_equal(back, host)
        codec.close()
        del d_src, d_out
        comp[r - 2] ^= 0x04   # the footer
        e = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
        assert _lib.is_error(e) and "match checksum" in _lib.error_name(e), _lib.error_name(e)
        comp[r - 2] ^= 0x04
        comp[r // 2] ^= 0x01  # inside a block of the third run: that run's error is the frame's, without a detour over the general kernel
        t0 = time.time()
        e = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r)
        assert _lib.is_error(e) and time.time() - t0 < 60
    finally:
        L.zl_freeDCtx(dctx)


def test_no_sequence_reaches_in_front_of_its_section():
    """Sections of 64 MiB are self-contained (include/zl_hip.h): the far search never looks across them, and neither may the
    one-byte backward extension of the stride-2 body -- which checked the bytes in front of the chunk in the FRAME, not in the
    section, until round 3: a match whose source is a section's first byte could be extended onto the byte in front of the
    section.  (A pledged stream's later jobs hold their own section only: the read faulted; scripts/fuzz_stream.py.)  The input
    makes exactly that case -- the last byte of section 1 and the first 64 bytes of section 2 appear again, 1000 bytes into
    section 2, on an even position, among random bytes -- and the frame is decoded section by section on two device slots, each
    of which has nothing but its own section: both slots must be used and the bytes must come back."""
    from zstdlite_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(77)
    sec = 64 << 20
    host = rng.integers(0, 256, sec + (1 << 20), dtype=np.uint8)
    host[sec - 1] = ord("Q")
    head = host[sec:sec + 64].copy()
    host[sec + 999] = ord("Q")
    host[sec + 1000:sec + 1064] = head
    n = host.size
    cap = L.zl_compressBound(n)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    try:
        L.zl_CCtx_setParameter(cctx, 9003, 2); L.zl_CCtx_setParameter(cctx, 9006, 1)  # one frame, seek table
        assert L.zl_DCtx_setParameter(dctx, 9008, 2) == 0                               # two device slots
        buf = np.empty(cap, dtype=np.uint8)
        r = L.zl_compress2(cctx, buf.ctypes.data, cap, host.ctypes.data, n)
        assert not _lib.is_error(r), _lib.error_name(r)
        back = np.zeros(n, dtype=np.uint8)
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, buf.ctypes.data, r)
        assert r2 == n, _lib.error_name(r2)
        assert np.array_equal(back, host)
        assert int(L.zl_dctx_last_slots(dctx)) == 2  # (a section that needed the one in front of it would have sent the call back to one slot)
        # ... and as a pledged stream, whose second job has only section 2 on the device
        import zstdlite_amd as z
        from zstdlite_amd import stream
        sc = z.zstd_cctx()
        comp = stream.Compressor(sc._ptr, pledged=n)
        raw = host.tobytes()
        out = [comp.feed(raw[p:p + (8 << 20)]) for p in range(0, n, 8 << 20)]
        out.append(comp.finish())
        assert b"".join(out) == z.zstd_compress(raw)
    finally:
        L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)


def _ref_batch_lib():
    import os
    from zl_testlib import REF_SO
    R = C.CDLL(os.path.join(os.path.dirname(REF_SO), "libzl_ref_batch.so"))
    assert R.zlr_init(REF_SO.encode()) == 0
    R.zlr_batch.restype = C.c_double
    R.zlr_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return R


@pytest.mark.parametrize("kind", ["rand", "iid10", "text", "int20"])
def test_every_input_at_one_gib(codec, kind):
    """Every SURVEY 8(d) input on its own at 2^30 bytes; kind "rand" is BASELINE configs[2] read literally: zstd_compress() of
    a 1 GiB vector of uniform random bytes, level 3 -- 8192 frames of one raw block each, inside the bound, back bit for bit."""
    dev = torch.device("cuda", 0)
    n = 1 << 30
    src = gen_device(kind, n, dev)
    comp, total, fs = codec.compress(src)
    assert len(fs) == n // CHUNK and total == int(fs.sum()) and total <= codec.bound(n)
    if kind == "rand":
        # a raw block: frame header (magic 4, descriptor 1, content size 4) + block header 3 + the bytes
        assert (fs == CHUNK + 12).all(), (int(fs.min()), int(fs.max()))
        head = comp[:12].cpu().numpy()
        assert bytes(head[:4]) == b"\x28\xb5\x2f\xfd" and (int(head[9]) & 6) == 0  # Raw_Block
    else:
        ref_ratio = {"iid10": 2.042, "text": 4.212, "int20": 1.272}[kind]  # BASELINE.md section 2, reference level 3
        assert n / total >= ref_ratio / 1.05, (kind, n / total)
    back = torch.empty_like(src)
    got, res = codec.decompress(comp, codec.frame_table(fs, codec.chunk_content_sizes(n)), back)
    assert got == n and (res == CHUNK).all() and torch.equal(back, src)


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref")
def test_one_gib_of_frames_through_the_reference_decoder(codec):
    """The encoder's whole 1 GiB output -- 8192 frames -- is decoded by the REFERENCE's ZSTD_decompressDCtx, one context per host
    thread (oracle/zl_ref_batch.c), and must regenerate the vector bit for bit: until round 4 the largest input that went to the
    reference decoder was 320 MiB."""
    import os
    dev = torch.device("cuda", 0)
    n = 1 << 30
    src, _, _ = gen_mixed(n, dev)
    comp, total, fs = codec.compress(src)
    blob = comp[:total].cpu().numpy()
    host = src.cpu().numpy()
    k = len(fs)
    in_size = np.asarray(fs, dtype=np.uint64)
    in_off = np.concatenate([[0], np.cumsum(in_size)[:-1]]).astype(np.uint64)
    out_cap = np.full(k, CHUNK, dtype=np.uint64)
    out_off = (np.arange(k, dtype=np.uint64) * np.uint64(CHUNK))
    out = np.zeros(n, dtype=np.uint8)
    out_size = np.zeros(k, dtype=np.uint64)
    R = _ref_batch_lib()
    R.zlr_batch(1, 3, min(os.cpu_count() or 1, 256), None, 0, k, blob.ctypes.data, in_off.ctypes.data, in_size.ctypes.data,
                out.ctypes.data, out_off.ctypes.data, out_cap.ctypes.data, out_size.ctypes.data)
    assert (out_size == CHUNK).all(), out_size[out_size != CHUNK][:4]
    assert np.array_equal(out, host)


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref")
def test_device_one_frame_with_checksum_is_complete_when_async_returns():
    """ONE frame with a content checksum through the device-resident entry: when zl_compress_device_async returns
    the frame in d_dst is complete and d_src may be overwritten -- the checksum is the one of the bytes that were compressed,
    whatever happens to d_src before zl_compress_device_finish."""
    from zl_testlib import Ref
    dev = torch.device("cuda", 0)
    n = 5 * CHUNK + 777
    c = DeviceCodec(include_checksum=True, frame_layout="one_frame")
    src = gen_device("text", n, dev)
    keep = src.cpu().numpy().tobytes()
    dst = torch.zeros(c.bound(n), dtype=torch.uint8, device=dev)
    r = c.L.zl_compress_device_async(c.cctx, C.c_void_p(dst.data_ptr()), dst.numel(), C.c_void_p(src.data_ptr()), n, None)
    assert r == 0
    src.zero_()  # the caller reuses its input before _finish
    torch.cuda.synchronize()
    sizes = (C.c_size_t * c.L.zl_compress_device_nframes(n))()
    total = c.L.zl_compress_device_finish(c.cctx, sizes)
    assert total < (1 << 62) and total == sum(sizes)
    frame = dst[:total].cpu().numpy().tobytes()
    assert (frame[4] >> 2) & 1  # Content_Checksum_flag
    ref = Ref()
    assert ref.decompress(frame, n) == keep  # (the reference verifies the checksum)
    c.close()
