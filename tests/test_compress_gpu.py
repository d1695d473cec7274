"""Parity tests for the HIP encoder, through the C-ABI.  The bar (BASELINE.json north_star):
output is valid zstd that the REFERENCE decoder (oracle/_ref) and the oracle restatement
round-trip bit-exact, it is deterministic, and the level-3 ratio stays within 5 % of the
reference libzstd on the SURVEY 8(d) inputs.  Mirrors tests/testthat/test-compress-raw.R,
test-round-trip.R, test-cctx.R, test-checksums.R."""
import os

import numpy as np
import pytest

import zstdlite_amd as z
from zl_testlib import GENERATORS, have_ref

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1000, 4096, 65535, 65536, 131071, 131072, 131073, 300000, 1 << 20]


def _decoders(oracle):
    decs = [("oracle", lambda c, n: oracle.decompress(c, cap=n))]
    if have_ref():
        from zl_testlib import Ref
        r = Ref()
        decs.append(("reference", lambda c, n: r.decompress(c, n)))
    return decs


@pytest.mark.parametrize("gen", ["text", "iid10", "int20", "rand"])
def test_roundtrip_through_reference_decoder(gen, oracle):
    decs = _decoders(oracle)
    cctx = z.zstd_cctx()
    for n in SIZES:
        data = GENERATORS[gen](n, seed=n + 17)
        comp = z.zstd_compress(data, cctx=cctx)
        assert len(comp) <= z.zstd_compress_bound(n)
        for name, dec in decs:
            assert dec(comp, n) == data, (gen, n, name)
        assert z.zstd_decompress(comp) == data, (gen, n, "hip")  # identical(dat, zstd_decompress(zstd_compress(dat)))


def test_special_inputs(oracle):
    decs = _decoders(oracle)
    rng = np.random.default_rng(1)
    cases = {
        "zeros": bytes(300000),
        "one_byte_runs": b"".join(bytes([i % 251]) * (i % 97 + 1) for i in range(4000)),
        "period3": b"abc" * 70000,
        "period257": bytes(range(256)) * 3 + b"x" + (bytes(range(256)) + b"y") * 500,
        "ramp_u16": np.arange(100000, dtype="<u2").tobytes(),
        "sparse": bytes(np.where(rng.random(250000) < 0.02, rng.integers(0, 256, 250000), 0).astype(np.uint8)),
        "text_then_rand": GENERATORS["text"](200000) + GENERATORS["rand"](200000),
        "two_symbols": bytes(rng.integers(0, 2, 150000, dtype=np.uint8) * 255),
        "all_256_symbols_skewed": bytes((rng.zipf(1.3, 200000) % 256).astype(np.uint8)),
    }
    for name, data in cases.items():
        comp = z.zstd_compress(data)
        for dn, dec in decs:
            assert dec(comp, len(data)) == data, (name, dn)
        assert z.zstd_decompress(comp) == data, name


def test_deterministic_and_cctx_reuse():
    """tests/testthat/test-cctx.R:5-37: on-the-fly context == reused zstd_cctx(), twice."""
    data = GENERATORS["text"](500000)
    cctx = z.zstd_cctx()
    a = z.zstd_compress(data)
    assert a == z.zstd_compress(data, cctx=cctx)
    assert a == z.zstd_compress(data, cctx=cctx)
    assert z.zstd_serialize({"a": list(range(1000))}) == z.zstd_serialize({"a": list(range(1000))}, cctx=cctx)


def test_deterministic_when_many_positions_share_a_bucket(oracle):
    """The match tables are filled with plain 16-bit LDS stores, one wave after the other; lanes of a
    wave that hit the same bucket are arbitrated by the hardware.  Inputs that maximise such
    conflicts (short periods, runs, a tiny alphabet) must still compress to the same bytes every
    time, with a fresh and with a reused context, and round-trip."""
    import numpy as np
    rng = np.random.default_rng(5)
    cases = [bytes([1, 2, 3]) * 90000, b"ab" * 150000, b"\x00" * 300000,
             bytes(rng.integers(0, 2, 400000, dtype=np.uint8)),
             (b"abcdefgh" * 40 + bytes(rng.integers(0, 256, 64, dtype=np.uint8))) * 700]
    cctx = z.zstd_cctx()
    for data in cases:
        first = z.zstd_compress(data)
        for _ in range(3):
            assert z.zstd_compress(data) == first
            assert z.zstd_compress(data, cctx=cctx) == first
        assert oracle.decompress(first, cap=len(data)) == data
        assert z.zstd_decompress(first) == data


def test_checksum_adds_four_bytes_per_frame():
    """tests/testthat/test-checksums.R:2-35."""
    dat = list(range(1, 11))
    v1 = z.zstd_serialize(dat)
    cctx = z.zstd_cctx(include_checksum=True)
    v2 = z.zstd_serialize(dat, cctx=cctx)
    assert len(v1) + 4 == len(v2)
    assert z.zstd_unserialize(v2) == dat
    bad = bytearray(v2)
    bad[-1] = (bad[-1] + 1) & 0xFF
    with pytest.raises(z.ZstdError, match="doesn't match checksum"):
        z.zstd_unserialize(bytes(bad))
    assert z.zstd_unserialize(bytes(bad), dctx=z.zstd_dctx(validate_checksum=False)) == dat
    # a checksum context writes ONE frame like every other host call (zstd.c:27733: one XXH64 over the whole content): four
    # bytes longer than without, and the flag in the frame header is the only other difference.  The independent-frames layout
    # (on request) checks each 128 KiB frame by itself.
    big = GENERATORS["text"](400000)
    one, onec = z.zstd_compress(big), z.zstd_compress(big, include_checksum=True)
    assert len(onec) == len(one) + 4 and onec[:4] == one[:4] and onec[4] == one[4] | 4 and onec[5:-4] == one[5:]
    assert z.zstd_info(onec)["compressed_size"] == len(onec) and z.zstd_info(onec)["has_checksum"]
    assert z.zstd_decompress(onec) == big
    assert len(z.zstd_compress(big, include_checksum=True, frame_layout="frames")) == len(z.zstd_compress(big, frame_layout="frames")) + 4 * 4


def test_serialize_roundtrip():
    """tests/testthat/test-round-trip.R:3-44 (pickle stands in for R's serialize)."""
    obj = {"mtcars": [[21.0, 6, 160.0], [22.8, 4, 108.0]], "s": "x" * 1000, "v": np.arange(100000).tolist()}
    assert z.zstd_unserialize(z.zstd_serialize(obj)) == obj
    assert z.zstd_unserialize(z.zstd_serialize(obj, level=5, num_threads=2)) == obj


def test_batch_compress(oracle):
    items = [GENERATORS[g](n, seed=i) for i, (g, n) in enumerate(
        [("text", 0), ("text", 300), ("iid10", 5000), ("rand", 777), ("int20", 70000), ("text", 200000), ("text", 131072)])]
    comps = z.zstd_compress_batch(items)
    singles = [z.zstd_compress(x, frame_layout="frames") for x in items]  # batch items are independent frames per chunk
    # (round 5: ONE input of 16 KiB - 16 MiB at levels 3 - 5 is searched by the deep match finder, which sorts one contiguous input;
    # the objects of a batch keep the table parse.  Same bytes where both take the tables: small objects, and every size at level 1)
    small = lambda x: len(x) <= 16384
    assert [c for x, c in zip(items, comps) if small(x)] == [c for x, c in zip(items, singles) if small(x)]
    assert [z.zstd_compress(x) for x in items if small(x)] == [c for x, c in zip(items, comps) if small(x)]
    for x, c, s in zip(items, comps, singles):
        assert len(c) <= len(s) * 1.10 + 16, (len(x), len(c), len(s))
    fast = z.zstd_compress_batch(items, level=1)
    assert fast == [z.zstd_compress(x, level=1, frame_layout="frames") for x in items]
    assert [z.zstd_compress(x, level=1) for x in items if len(x) <= 131072] == [c for x, c in zip(items, fast) if len(x) <= 131072]
    for x, c in zip(items, fast):
        assert oracle.decompress(c, cap=len(x)) == x
    for x, c in zip(items, comps):
        assert oracle.decompress(c, cap=len(x)) == x
    assert z.zstd_decompress_batch([c for x, c in zip(items, comps) if len(x) <= 131072]) == [x for x in items if len(x) <= 131072]


@pytest.mark.parametrize("gen,slack", [("text", 1.05), ("iid10", 1.05), ("int20", 1.05), ("rand", 1.001)])
def test_ratio_within_five_percent_of_reference(gen, slack, ref):
    """north_star: zstd_compress() level-3 ratio within 5 % of CPU libzstd (SURVEY 8d inputs)."""
    n = 8 << 20
    data = GENERATORS[gen](n)
    ours = len(z.zstd_compress(data))
    theirs = len(ref.compress(data, level=3))
    assert ours <= theirs * slack, (gen, ours, theirs, ours / theirs)


@pytest.mark.parametrize("level", [1, 2])
def test_lower_levels_track_the_reference_at_the_same_level(level, ref, oracle):
    """Levels below 3 select the faster operating point of the encoder (every other position probed);
    the ratio must stay within 5 % of the reference library at that level, and frames must round-trip."""
    n = 8 << 20
    for gen in ("text", "iid10", "int20"):
        data = GENERATORS[gen](n)
        comp = z.zstd_compress(data, level=level)
        theirs = len(ref.compress(data, level=level))
        assert len(comp) <= theirs * 1.05, (gen, level, len(comp), theirs)
        assert z.zstd_decompress(comp) == data
        if gen == "text":
            assert comp != z.zstd_compress(data, level=3)
    small = GENERATORS["text"](70000, seed=3)
    assert oracle.decompress(z.zstd_compress(small, level=level), cap=len(small)) == small


def test_dictionary_compression_of_small_objects(ref, oracle):
    """BASELINE config C4's shape (vignettes/dictionaries.Rmd:95-141): many small objects, one trained
    dictionary.  The dictionary's content is history in front of every frame: frames name its id,
    decode with the reference library, the oracle and the GPU decoder given the dictionary, fail
    without it, and are clearly smaller than without a dictionary.  Frames of a single block also take
    the dictionary's entropy tables (treeless literals, FSE repeat modes from the first block) when
    that is cheaper, which puts the ratio level with the reference's."""
    import pickle
    rng = np.random.default_rng(3)
    names = [f"name_{i:02d}_{'x' * (i % 7)}" for i in range(50)]

    def obj():
        perm = rng.permutation(50)[: int(rng.integers(10, 40))]
        return pickle.dumps({names[j]: int(rng.integers(0, 1000)) for j in perm}, protocol=4)
    items = [obj() for _ in range(3000)]
    d = ref.train_dict(items[:1500], 5000)
    cctx = z.zstd_cctx(dict=d)
    plain = z.zstd_compress_batch(items)
    withd = z.zstd_compress_batch(items, cctx=cctx)
    assert withd == z.zstd_compress_batch(items, cctx=cctx)  # deterministic, context reusable
    assert [z.zstd_compress(x, cctx=cctx) for x in items[:20]] == withd[:20]
    raw = sum(map(len, items))
    assert sum(map(len, withd)) < 0.7 * sum(map(len, plain))
    cpu = sum(len(ref.compress(x, level=3, dict=d)) for x in items[:500])
    ours = sum(map(len, withd[:500]))
    assert ours <= cpu * 1.03, (ours, cpu)
    # the first block of most of these frames has treeless literals (Literals_Block_Type 3): only legal with a dictionary
    # (magic, descriptor, 4-byte dictionary id, 1- or 2-byte content size, 3-byte block header, then the literals section)
    assert z.zstd_info(withd[0])["dict_id"] > 65535
    treeless = sum(1 for x, f in zip(items, withd) if (f[5 + 4 + (1 if len(x) < 256 else 2) + 3] & 3) == 3)
    assert treeless > len(withd) // 2, treeless
    # inputs the dictionary's tables do not suit (bytes its Huffman table cannot code, other statistics)
    # fall back to their own tables or to raw blocks; every size class still round-trips
    odd = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in (1, 5, 6, 7, 63, 64, 300, 5000)]
    odd += [bytes(rng.integers(0, 4, int(k), dtype=np.uint8) + 250) for k in (6, 40, 64, 300, 5000, 20000)]
    odd += [items[3][:k] for k in (1, 6, 7, 30, 100)] + [items[5] * 3, items[7] + bytes(300), b"", bytes(70)]
    for x, f in zip(odd, z.zstd_compress_batch(odd, cctx=cctx)):
        assert ref.decompress(f, max(len(x), 1), dict=d) == x, len(x)
        assert oracle.decompress(f, cap=max(len(x), 1), dict=d) == x, len(x)
    assert z.zstd_info(withd[0])["dict_id"] == z.zstd_dict_id(d) != 0
    for i in range(0, len(items), 37):
        assert ref.decompress(withd[i], len(items[i]), dict=d) == items[i]
        assert oracle.decompress(withd[i], cap=len(items[i]), dict=d) == items[i]
    assert z.zstd_decompress_batch(withd, dict=d) == items
    with pytest.raises(z.ZstdError):
        z.zstd_decompress(withd[0])
    # a large input with a dictionary: ONE frame whose first chunk sees the dictionary (test_checksum_and_dictionary_contexts_write_one_frame);
    # in the independent-frames layout, on request, every 128 KiB frame sees it
    big = b"".join(items)
    comp = z.zstd_compress(big, cctx=z.zstd_cctx(dict=d, frame_layout="frames"))
    assert z.zstd_decompress(comp, dctx=z.zstd_dctx(dict=d)) == big
    assert ref.decompress(comp[: z.zstd_info(comp)["compressed_size"]], 131072, dict=d) == big[:131072]
    # a raw-content dictionary (no magic): no id in the frame, still history
    rawd = b"".join(items[:10])
    c2 = z.zstd_compress(items[20], cctx=z.zstd_cctx(dict=rawd))
    assert z.zstd_info(c2)["dict_id"] == 0
    assert ref.decompress(c2, len(items[20]), dict=rawd) == items[20]


# ---- the one-frame (windowed) layout: zl_compress2's default for inputs above one chunk --------------------------
@pytest.mark.parametrize("n", [131073, 262144, (2 << 20) - 1, 2 << 20, (2 << 20) + 1, 5000000])
def test_one_frame_is_one_standard_frame(n, ref):
    """What zstd_compress() writes is ONE zstd frame, as ZSTD_compress2 does (src/raw-file.c:74): the first-frame
    helpers of the reference's read side (src/raw-file.c:150-155) see all of it, the reference decoder regenerates
    it, and so do we.  Sizes around the 2 MiB window: Single_Segment up to it, a Window_Descriptor above."""
    data = (GENERATORS["text"](n // 2, seed=n) + GENERATORS["iid10"](n - n // 2, seed=n + 1))
    comp = z.zstd_compress(data)
    assert ref.L.ZSTD_findFrameCompressedSize(comp, len(comp)) == len(comp)
    assert ref.L.ZSTD_getFrameContentSize(comp, len(comp)) == n
    info = z.zstd_info(comp)
    assert info["compressed_size"] == len(comp) and info["uncompressed_size"] == n
    assert ref.decompress(comp, n) == data
    assert z.zstd_decompress(comp) == data
    assert z.zstd_compress(data) == comp  # deterministic


@pytest.mark.parametrize("n", [131073, 5000000])
def test_checksum_and_dictionary_contexts_write_one_frame(n, ref, oracle):
    """zstd_compress(x, include_checksum=TRUE) and zstd_compress(x, dict=) of an input above 128 KiB used to
    return one frame per chunk, of which the reference's reader (first frame only, src/raw-file.c:150-155) silently took the
    first.  They write ONE frame now: the reference's first-frame helper covers all of it, the reference decoder verifies the
    checksum (zstd.c:27733: XXH64 over the whole content) and finds the dictionary's content in front of the frame
    (zstd.c:27550).  (320 MiB: tests/test_fullsize_gpu.py.)"""
    import pickle
    rng = np.random.default_rng(n)
    data = GENERATORS["text"](n // 2, seed=n) + GENERATORS["iid10"](n - n // 2, seed=n + 1)
    # ---- checksum
    comp = z.zstd_compress(data, include_checksum=True)
    assert ref.L.ZSTD_findFrameCompressedSize(comp, len(comp)) == len(comp) and ref.L.ZSTD_getFrameContentSize(comp, len(comp)) == n
    assert z.zstd_info(comp)["has_checksum"]
    assert ref.decompress(comp, n) == data and oracle.decompress(comp, cap=n) == data and z.zstd_decompress(comp) == data
    for at in (len(comp) - 1, len(comp) - 4):  # the stored checksum: both decoders must notice
        bad = bytearray(comp)
        bad[at] ^= 0x10
        r, _ = ref.decompress_code(bytes(bad), n)
        assert ref.error_name(r) == "Restored data doesn't match checksum"
        with pytest.raises(z.ZstdError, match="doesn't match checksum"):
            z.zstd_decompress(bytes(bad))
        assert z.zstd_decompress(bytes(bad), validate_checksum=False) == data
    assert z.zstd_compress(data, include_checksum=True, seek_table=True)[:len(comp)] == comp
    assert z.zstd_decompress(z.zstd_compress(data, include_checksum=True, seek_table=True)) == data
    # ---- dictionary: objects like the dictionary's samples, many of them in one buffer
    names = [f"name_{i:02d}_{'x' * (i % 7)}" for i in range(50)]

    def obj():
        perm = rng.permutation(50)[: int(rng.integers(10, 40))]
        return pickle.dumps({names[j]: int(rng.integers(0, 1000)) for j in perm}, protocol=4)
    items = [obj() for _ in range(1500)]
    d = ref.train_dict(items, 5000)
    body = b""
    while len(body) < n:
        body += obj()
    body = body[:n]
    for dic, did in ((d, z.zstd_dict_id(d)), (b"".join(items[:10]), 0)):  # a trained dictionary; raw content (no id in the frame)
        cd = z.zstd_compress(body, dict=dic, include_checksum=True)
        assert ref.L.ZSTD_findFrameCompressedSize(cd, len(cd)) == len(cd) and ref.L.ZSTD_getFrameContentSize(cd, len(cd)) == n
        assert z.zstd_info(cd)["dict_id"] == did
        assert ref.decompress(cd, n, dict=dic) == body and oracle.decompress(cd, cap=n, dict=dic) == body
        assert z.zstd_decompress(cd, dict=dic) == body
        assert len(cd) < len(z.zstd_compress(body, include_checksum=True)) * 1.05  # (the first chunk is searched with the dictionary's smaller tables)
        if did:
            with pytest.raises(z.ZstdError):
                z.zstd_decompress(cd)
    # a reference-made large frame with the dictionary (its first block may ride on the dictionary's tables): bit-exact too
    rc = ref.compress(body, level=3, dict=d, checksum=True)
    assert z.zstd_decompress(rc, dict=d) == body


@pytest.mark.parametrize("name,n", [("records", 16 << 20), ("corpus", None)])
def test_ratio_within_five_percent_on_long_range_inputs(name, n, ref):
    """north_star's ratio bar on inputs whose redundancy lies further back than one 128 KiB chunk (VERDICT r1:
    the four SURVEY inputs hold nothing beyond a few hundred bytes): repeats at 1 MiB distance, and real text."""
    from zl_testlib import LONG_RANGE
    data = LONG_RANGE[name](n)
    comp = z.zstd_compress(data)
    theirs = len(ref.compress(data, level=3))
    assert ref.decompress(comp, len(data)) == data
    assert z.zstd_decompress(comp) == data
    assert len(comp) <= theirs * 1.05, (name, len(data) / len(comp), len(data) / theirs)


def test_frame_layout_option_and_device_hint(ref):
    import torch
    from zstdlite_amd.device import DeviceCodec
    from zl_testlib import LONG_RANGE
    data = LONG_RANGE["records"](3 << 20) + GENERATORS["text"](1 << 20)
    frames = z.zstd_compress(data, frame_layout="frames")
    one = z.zstd_compress(data, frame_layout="one_frame")
    assert ref.L.ZSTD_findFrameCompressedSize(frames, len(frames)) < len(frames)  # the first of many
    assert z.zstd_decompress(frames) == data and z.zstd_decompress(one) == data
    assert len(one) < len(frames) / 2  # only the one-frame layout sees the 1 MiB repeats
    # device entries: independent frames by default; one frame on request, decoded with and without the chunk hint
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    codec = DeviceCodec(frame_layout="one_frame")
    dst, total, sizes = codec.compress(src)
    assert bytes(dst[:total].cpu().numpy()) == one
    back = torch.empty_like(src)
    tbl = codec.frame_table([total], [len(data)])
    for hint in (sizes, None, sizes[::-1].copy(), sizes[:-1]):  # a hint that does not fit the frame is ignored
        back.zero_()
        got, _ = codec.decompress(dst, tbl, back, chunk_sizes=hint)
        assert got == len(data) and torch.equal(back, src)


def test_c1_mtcars_head_shape_on_the_gpu(ref, oracle):
    """Config C1's input (tests/golden/make_c1.py) through the HIP path: the same frame header as the README shows for
    the reference (README.md:102: 28 b5 2f fd 60 e8 02), both CPU decoders regenerate the 1000 bytes, and the size is within
    north_star's 5 % of the reference's on the same bytes (test_ratio_sweep_small_sizes pins 1 KB to 1 MB)."""
    import os
    from zl_testlib import GOLDEN
    buf = open(os.path.join(GOLDEN, "c1_mtcars_head.bin"), "rb").read()
    comp = z.zstd_compress(buf)
    assert comp[:7] == bytes.fromhex("28b52ffd60e802")
    assert len(comp) <= len(ref.compress(buf, level=3)) * 1.05  # (round 2: 1.127; inputs of at most 16 KiB hash four bytes and see their own window since round 3)
    assert ref.decompress(comp, 1000) == buf and oracle.decompress(comp) == buf and z.zstd_decompress(comp) == buf


@pytest.mark.parametrize("layout", ["auto", "frames"])
def test_seek_table_is_skippable_and_spares_the_walk(layout, ref):
    """the optional seek table is a skippable frame (zstd.h:138) behind the data.  The reference decoder
    skips it (zstd.c:41722-41733), a first-frame reader never sees it, our decoder reads the chunk sizes from it -- and
    falls back to the walk when the table does not describe the data."""
    import ctypes as C
    from zstdlite_amd import _lib
    data = GENERATORS["text"](700000, seed=5) + GENERATORS["iid10"](500000, seed=6) + GENERATORS["rand"](200000, seed=7)
    plain = z.zstd_compress(data, frame_layout=layout)
    comp = z.zstd_compress(data, frame_layout=layout, seek_table=True)
    nchunks = -(-len(data) // 131072)
    # one frame: the private trailer; independent frames: zstd's seekable format (zl_format.h)
    assert comp[:len(plain)] == plain and len(comp) == len(plain) + (8 + 4 * nchunks + 20 if layout == "auto" else 8 + 8 * nchunks + 9)
    assert int.from_bytes(comp[len(plain):len(plain) + 4], "little") == 0x184D2A5E
    if layout == "frames":
        assert comp[-4:] == (0x8F92EAB1).to_bytes(4, "little") and comp[-5] == 0 and int.from_bytes(comp[-9:-5], "little") == nchunks
        ent = comp[len(plain) + 8:-9]
        assert sum(int.from_bytes(ent[8 * i:8 * i + 4], "little") for i in range(nchunks)) == len(plain)
        assert sum(int.from_bytes(ent[8 * i + 4:8 * i + 8], "little") for i in range(nchunks)) == len(data)
    assert ref.decompress(comp, len(data)) == data           # the whole buffer: the table is skipped
    assert z.zstd_decompress(comp) == data                   # through the table
    L = _lib.lib()
    frames = (_lib.FrameT * (nchunks + 1))()
    cnt = L.zl_index_frames(comp, len(comp), frames, nchunks + 1)
    assert cnt == (1 if layout == "auto" else nchunks)
    assert sum(frames[i].dst_cap for i in range(cnt)) == len(data)
    # a table that lies about one size still sums up only if another lies too: both chunks then fail to decode from where
    # the table says they start, and the decoder goes back to walking the data
    bad = bytearray(comp)
    t0, step = len(plain) + 8, (4 if layout == "auto" else 8)
    a = int.from_bytes(bad[t0:t0 + 4], "little"); b = int.from_bytes(bad[t0 + step:t0 + step + 4], "little")
    bad[t0:t0 + 4] = (a + 1).to_bytes(4, "little"); bad[t0 + step:t0 + step + 4] = (b - 1).to_bytes(4, "little")
    assert z.zstd_decompress(bytes(bad)) == data


def test_level_five_is_a_stronger_operating_point(ref):
    """Levels >= 5 run the encoder's operating point above level 3 (include/zl_hip.h): in the independent-frames layout match
    tables of 2^14 + 2^13 entries and verification up to 72 bytes per lane (the one-frame layout has those at every level:
    its callers sit behind PCIe), in the one-frame layout one position in four in the far search instead of one in eight.
    The lazy / optimal parsers behind the reference's levels >= 5 are not built, so the bar is what the knob does: clearly
    smaller output than level 3 in BOTH layouts, level with the reference's level 3 on real text, valid for both decoders."""
    import warnings
    from zl_testlib import LONG_RANGE
    data = LONG_RANGE["corpus"](None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        l3, l5 = z.zstd_compress(data, level=3), z.zstd_compress(data, level=5)
        f3, f5 = z.zstd_compress(data, level=3, frame_layout="frames"), z.zstd_compress(data, level=5, frame_layout="frames")
    assert len(l5) < len(l3) * 0.985, (len(l3), len(l5))
    assert len(f5) < len(f3) * 0.99, (len(f3), len(f5))  # (0.96 until round 5 gave levels 3 - 5 the deep match finder on inputs of this size: level 3 came down to it)
    assert len(l5) <= len(ref.compress(data, level=3)) * 1.01
    assert ref.decompress(l5, len(data)) == data and z.zstd_decompress(l5) == data
    assert z.zstd_decompress(f5) == data
    text = GENERATORS["text"](4 << 20, seed=9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t3, t5 = z.zstd_compress(text, level=3, frame_layout="frames"), z.zstd_compress(text, level=5, frame_layout="frames")
    assert len(t5) < len(t3) * 0.985 and z.zstd_decompress(t5) == text
    first = z.zstd_info(t5)["compressed_size"]  # the first of the independent frames, through the reference decoder
    assert ref.decompress(t5[:first], 131072) == text[:131072]


@pytest.mark.parametrize("level", [6, 7, 9, 12, 15])
def test_levels_six_to_fifteen_track_the_reference_at_the_same_level(ref, level):
    """Levels 6 - 15 run the deep match finder (csrc/zl_deep.hip): every position's K most recent earlier
    occurrences inside the window -- the candidates the reference's hash chains / row hash visit (zstd.c:31546-33746), with the
    reference's own K, lazy depth and window at that level --, the one- and two-step lazy rule, and the entropy stage that follows
    the decoder's three-entry repeat-offset history (zstd.c:19665, :44291).  The REFERENCE decoder must regenerate every input bit
    for bit, in both layouts, from one block to many sections; and the bar of north_star holds AT THE SAME LEVEL: compressed size
    at most 1.05x the reference's on real text (the python sources: 1.085x / 1.128x / 1.149x at 6 / 9 / 12 until round 4), on the
    synthetic text and on the long-range records."""
    import warnings
    from zl_testlib import LONG_RANGE
    from ratio_inputs import INPUTS
    corpus = LONG_RANGE["corpus"](None)
    text = GENERATORS["text"](5 << 20, seed=level)
    records = LONG_RANGE["records"](8 << 20)
    cases = [("corpus", corpus), ("corpus_3M", corpus[:3 << 20]), ("text", text), ("records", records),
             ("iid10", GENERATORS["iid10"](1 << 20, seed=level)), ("one_chunk", corpus[:100000]), ("tiny", corpus[:900]), ("seven", b"abcdefg"), ("empty", b"")]
    cases += [(name, gen(1 << 18)) for name, gen in list(INPUTS.items())[:8]]
    outs = {}
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # (every one of these levels is an operating point of its own: the mirror has nothing to say)
        for name, data in cases:
            for layout in ("auto", "frames"):
                out = z.zstd_compress(data, level=level, frame_layout=layout)
                outs[name, layout] = out
                if layout == "auto":
                    assert ref.decompress(out, len(data)) == data, (name, layout)
                else:  # independent frames: the reference decodes them one after the other
                    pos, got = 0, []
                    while pos < len(out):
                        k = z.zstd_info(out[pos:])["compressed_size"]
                        got.append(ref.decompress(out[pos:pos + k], 131072))
                        pos += k
                    assert b"".join(got) == data, (name, layout)
                assert z.zstd_decompress(out) == data, (name, layout)
        c5 = z.zstd_compress(corpus, level=5)
    for name, data in (("corpus", corpus), ("text", text), ("records", records)):
        ours, theirs = len(outs[name, "auto"]), len(ref.compress(data, level=level))
        assert ours <= theirs * 1.05, (name, level, ours, theirs, ours / theirs)
    # (level 5 on an input of this size is the deep match finder six candidates deep since round 5: levels 6 and 7 sit right below it)
    assert len(outs["corpus", "auto"]) < len(c5) * (1.0 if level <= 7 else 0.97)
    assert z.zstd_compress(corpus, level=level) == outs["corpus", "auto"]  # deterministic
    assert z.zstd_cctx_settings(z.zstd_cctx(level=level))["level"] == level


@pytest.mark.parametrize("level", [16, 19, 22])
def test_levels_sixteen_to_twenty_two_price_their_parse(ref, level):
    """Levels 16 - 22: the price-based parse of csrc/zl_opt.hip over the deep match finder's candidates -- every
    position's nearest offset per length, a forward pass over arrival prices from the chunk's own symbol statistics, the path read off
    backwards (the reference's btopt / btultra / btultra2, zstd.c:34586-36163).  The REFERENCE decoder must regenerate every input bit for
    bit, in both layouts, from nothing to many chunks; the output is deterministic; and the bar of north_star holds AT THE SAME LEVEL
    on real text (the python sources) and on the long-range records: at most 1.05x the reference's size (level 15's parse at these
    levels, round 5: 1.07 - 1.11x).  The synthetic text -- random words of a small vocabulary, every one with thousands of twins -- wants
    more than the 256 most recent twins of FOUR bytes (the reference's binary tree looks at all of them): a second search pass over an
    eight-byte hash is merged into the chains (1.09 - 1.10x -> 1.04 - 1.05x); held to 1.08x, and said so in DESIGN 3.3."""
    import warnings
    from zl_testlib import LONG_RANGE
    from ratio_inputs import INPUTS
    corpus = LONG_RANGE["corpus"](None)
    text = GENERATORS["text"](3 << 20, seed=level)
    records = LONG_RANGE["records"](8 << 20)
    cases = [("corpus", corpus), ("text", text), ("records", records), ("iid10", GENERATORS["iid10"](1 << 20, seed=level)), ("int20", GENERATORS["int20"](1 << 20, seed=level)),
             ("rand", GENERATORS["rand"](300000, seed=level)), ("runs", b"a" * 200000 + b"ab" * 100000 + bytes(1000)), ("one_chunk", corpus[:100000]),
             ("chunk_and_a_byte", corpus[:131073]), ("tiny", corpus[:900]), ("fifteen", corpus[:15]), ("seven", b"abcdefg"), ("empty", b"")]
    cases += [(name, gen(1 << 18)) for name, gen in list(INPUTS.items())[:8]]
    outs = {}
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # (an operating point of its own: the mirror has nothing to say)
        for name, data in cases:
            for layout in ("auto", "frames"):
                out = z.zstd_compress(data, level=level, frame_layout=layout)
                outs[name, layout] = out
                if layout == "auto":
                    assert ref.decompress(out, len(data)) == data, (name, layout)
                else:  # independent frames: the reference decodes them one after the other
                    pos, got = 0, []
                    while pos < len(out):
                        k = z.zstd_info(out[pos:])["compressed_size"]
                        got.append(ref.decompress(out[pos:pos + k], 131072))
                        pos += k
                    assert b"".join(got) == data, (name, layout)
                assert z.zstd_decompress(out) == data, (name, layout)
        c15 = z.zstd_compress(corpus, level=15)
    for name, data, bar in (("corpus", corpus, 1.05), ("records", records, 1.05), ("iid10", cases[3][1], 1.05), ("text", text, 1.08)):
        ours, theirs = len(outs[name, "auto"]), len(ref.compress(data, level=level))
        assert ours <= theirs * bar, (name, level, ours, theirs, ours / theirs)
    assert len(outs["corpus", "auto"]) < len(c15) * 0.96  # what the prices buy over level 15's lazy parse of the same candidates
    assert z.zstd_compress(corpus, level=level) == outs["corpus", "auto"]  # deterministic
    assert z.zstd_cctx_settings(z.zstd_cctx(level=level))["level"] == level


def test_negative_levels_store_their_literals_and_stay_valid(ref):
    """Negative levels (zstd.c:29721: targetLength = -level; zstd.c:20675: literal compression is off for `fast` with a target length, i.e.
    at every negative level): the frames carry Raw_Literals_Block sections only, decode through the reference, are larger than level 1's and
    report the level they were asked for.  (Their speed is on the bench line: `levels`.)"""
    import warnings
    data = GENERATORS["text"](3 << 20, seed=4) + GENERATORS["iid10"](1 << 20, seed=4)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        l1 = z.zstd_compress(data, level=1)
        for level in (-1, -5):
            out = z.zstd_compress(data, level=level)
            assert ref.decompress(out, len(data)) == data and z.zstd_decompress(out) == data
            assert len(out) > len(l1)
            assert z.zstd_cctx_settings(z.zstd_cctx(level=level))["level"] == level
            frames = z.zstd_compress(data, level=level, frame_layout="frames")
            # every block of the first frame: compressed blocks open with a literals section of type 0 (raw) or 1 (RLE), never Huffman
            fhd = frames[4]
            pos = 5 + (0 if (fhd >> 5) & 1 else 1) + {0: 1 if (fhd >> 5) & 1 else 0, 1: 2, 2: 4, 3: 8}[fhd >> 6]
            seen = 0
            while True:
                h = int.from_bytes(frames[pos:pos + 3], "little")
                last, btype, bsize = h & 1, (h >> 1) & 3, h >> 3
                if btype == 2:
                    assert frames[pos + 3] & 3 in (0, 1), (level, seen)
                    seen += 1
                pos += 3 + (1 if btype == 1 else bsize)
                if last:
                    break
            assert seen >= 1


def test_every_level_from_six_on_is_an_operating_point_of_its_own(ref):
    """zl_CCtx_getParameter reports 6 ... 15 as they were set, and a higher level never writes a larger frame than the one below it
    by more than noise (python sources, one frame)."""
    from zl_testlib import LONG_RANGE
    corpus = LONG_RANGE["corpus"](None)[: 6 << 20]
    sizes = [len(z.zstd_compress(corpus, level=lv)) for lv in range(5, 16)]
    for a, b in zip(sizes, sizes[1:]):
        assert b <= a * 1.003, sizes
    assert sizes[-1] < sizes[1] * 0.95, sizes
    assert len(set(sizes)) >= 9, sizes


def test_far_probe_gate_keeps_copies_of_random_bytes(ref):
    """Round 5: the far probe leaves a chunk alone when a sparse sample of its positions (one in 256, chosen by content) finds no
    tag further back -- chunks of random bytes, small integers, i.i.d. symbols, which were half of the kernel's time on the bench
    mix.  What must survive: random bytes that ARE a copy of something further back (no byte of them is cheap, every position has
    its twin) -- and the sizes of inputs whose far matches carry the ratio."""
    from zl_testlib import LONG_RANGE
    rng = np.random.default_rng(77)
    blob = bytes(rng.integers(0, 256, size=1 << 20, dtype=np.uint8))
    filler = bytes(rng.integers(0, 256, size=17 << 20, dtype=np.uint8))          # (above 16 MiB in all: the table parse and its far passes)
    data = blob + filler[: 600_000] + blob + filler[600_000:] + blob[: 300_000]   # copies 1.6 MiB and ~17 MiB back: the first inside the window
    out = z.zstd_compress(data)
    assert ref.decompress(out, len(data)) == data and z.zstd_decompress(out) == data
    assert len(out) < len(data) - (1 << 20) + (64 << 10), (len(out), len(data))   # the near copy costs next to nothing
    records = LONG_RANGE["records"](40 << 20)
    assert len(z.zstd_compress(records)) <= len(ref.compress(records, level=3)) * 1.05


def test_levels_three_to_five_at_the_edges_of_the_small_input_route(ref):
    """Levels 3 - 5 search ONE input of more than 16 KiB and at most 16 MiB with the deep match finder, everything else with the tables
    (DESIGN.md 3.3): both sides of both edges, in both layouts, through both decoders; the routes' sizes stay close to each other
    (an edge is not a cliff) and the python sources of a few hundred KB come out SMALLER than the reference's at the same level,
    which the table parse missed by 9 - 24 %."""
    from zl_testlib import LONG_RANGE
    text = GENERATORS["text"]((16 << 20) + 64, seed=41)
    for n in (16384, 16385, 16 << 20, (16 << 20) + 1):
        data = text[:n]
        for lv in (3, 5):
            for layout in ("auto", "frames"):
                out = z.zstd_compress(data, level=lv, frame_layout=layout)
                assert z.zstd_decompress(out) == data, (n, lv, layout)
                if layout == "auto":
                    assert ref.decompress(out, n) == data, (n, lv)
                    assert len(out) <= len(ref.compress(data, level=lv)) * 1.05, (n, lv)
    corpus = LONG_RANGE["corpus"](None)
    for n in (200_000, 1 << 20):
        data = corpus[3_000_000:3_000_000 + n]
        for lv in (3, 4, 5):
            assert len(z.zstd_compress(data, level=lv)) <= len(ref.compress(data, level=lv)) * 1.07, (n, lv)
        assert len(z.zstd_compress(data, level=3)) <= len(ref.compress(data, level=3)) * 0.99, n


def test_levels_sixteen_to_twenty_two_have_their_windows(ref):
    """Levels 16 ... 22: level 15's search and parse inside the reference's WINDOW at that level (zstd.c:29542-29548: 2^22 at 16, 2^23
    at 17 - 19, 2^25 at 20, 2^26 / 2^27 at 21 / 22 -- here at most a section, 2^26).  The deep match finder's sort hands a position
    its most recent twins whatever the distance, so the window costs no time; what it buys: a repeat 6 MiB back is invisible at 15 and
    found at 17, one 13 MiB back is found from 20 on.  The frames name their window, both decoders take them, and
    zl_CCtx_getParameter reports the level as given.  (Round 6: these levels also price their parse -- the test above.)"""
    import warnings
    a, b = GENERATORS["text"](2 << 20, seed=31), GENERATORS["text"](4 << 20, seed=32)
    near = a + b + a                                                   # the second copy lies 6 MiB behind the first
    far = a + GENERATORS["text"](11 << 20, seed=33) + a                # ... 13 MiB
    sizes = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, data, levels in (("near", near, (15, 16, 17, 19)), ("far", far, (19, 20, 22))):
            for lv in levels:
                out = z.zstd_compress(data, level=lv)
                assert ref.decompress(out, len(data)) == data and z.zstd_decompress(out) == data, (name, lv)
                sizes[name, lv] = len(out)
                assert z.zstd_cctx_settings(z.zstd_cctx(level=lv))["level"] == lv
    one_copy = len(z.zstd_compress(a, level=15))
    assert sizes["near", 16] > sizes["near", 15] - one_copy * 0.5      # (windows of 4 MiB both: the second copy is invisible to either; 16 prices its parse and is a few per cent smaller)
    assert sizes["near", 17] < sizes["near", 15] - one_copy * 0.9      # the second copy costs next to nothing inside 8 MiB
    assert sizes["near", 19] <= sizes["near", 17] * 1.001
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one_copy_19 = len(z.zstd_compress(a, level=19))                # (what the copy costs where it is NOT found, at that level's parse)
    assert sizes["far", 20] < sizes["far", 19] - one_copy_19 * 0.9
    assert sizes["far", 22] <= sizes["far", 20] * 1.001


def test_chunks_of_cheap_bytes_keep_only_long_matches(ref):
    """Round 4: a chunk whose bytes cost less than 3.6 bits and whose first 4 KiB hardly match keeps only new-offset
    matches of eight bytes and more (and, in one frame, far hints of twelve and more) -- on i.i.d. bytes over ten letters the short
    matches made the output larger and the search for them was all the kernel did.  The output must stay valid for the reference
    decoder, must not grow, and inputs that are cheap byte for byte BUT repetitive must keep the reference's ratio."""
    import numpy as np
    rng = np.random.default_rng(11)
    iid = GENERATORS["iid10"](6 << 20, seed=5)
    digits = bytes(rng.integers(48, 58, size=3 << 20, dtype=np.uint8))                      # 3.32 bits per byte, nothing repeats
    lines = [b",".join(b"%d" % v for v in rng.integers(0, 1000, size=12)) + b"\n" for _ in range(400)]
    table = b"".join(lines[i] for i in rng.integers(0, 400, size=60000))                     # cheap bytes AND whole lines that repeat
    head = bytes(rng.integers(48, 58, size=6000, dtype=np.uint8)) + GENERATORS["text"](2 << 20, seed=3)  # the probe sees no match, the rest is text
    # ("table": 400 different lines, so a line's last occurrence lies 19 KiB back on average -- further than the 2^13-entry LDS tables
    # of the table parse remember: 1.17x the reference's size in round 3, 1.21 - 1.25x in round 4.  Round 5: an input of this size is
    # searched by the deep match finder at level 3 too, which sees every occurrence inside the window: 1.016x)
    for name, data, bar in (("iid10", iid, 1.0), ("digits", digits, 1.0), ("table", table, 1.05), ("digits_then_text", head, 1.05)):
        theirs = len(ref.compress(data, level=3))
        for layout in ("auto", "frames"):
            out = z.zstd_compress(data, frame_layout=layout)
            assert z.zstd_decompress(out) == data, (name, layout)
            if layout == "auto":
                assert ref.decompress(out, len(data)) == data, name
                assert len(out) <= theirs * bar, (name, layout, len(out), theirs)
            elif name != "table":  # (independent 128 KiB frames cannot see a line's last occurrence further back: the long-range case)
                assert len(out) <= theirs * (bar + 0.01), (name, layout, len(out), theirs)
    assert len(iid) / len(z.zstd_compress(iid, frame_layout="frames")) >= 2.34  # (round 3, with the short matches: 2.344; entropy alone: 2.35)


def test_ratio_sweep_small_sizes(ref):
    """the 5 % ratio bar was pinned at 8 MiB and above only, while zstd_serialize() of ordinary R objects lives
    between 1 KB and 1 MB.  Eight kinds of input (tests/ratio_inputs.py: the SURVEY 8d generators, three kinds of float64
    vectors -- R's default numeric --, config C1's buffer repeated, a pickled data frame) at six sizes against
    ref.compress(level = 3): at most 1.05x its size wherever that holds, and where it does not (tests/golden/ratio_sweep_r03.json,
    the table scripts/ratio_sweep.py measured on MI355X: 1 KiB of a pickled frame or of rounded doubles; records that repeat at
    ratio 100 and more, where 20 bytes of fixed cost per block show) no more than 2 % above what the table recorded."""
    import json, os
    from ratio_inputs import INPUTS, SIZES
    from zl_testlib import GOLDEN
    recorded = json.load(open(os.path.join(GOLDEN, "ratio_sweep_r03.json")))
    over = {}
    for name, gen in INPUTS.items():
        for n in SIZES:
            data = gen(n)
            ours = z.zstd_compress(data)
            assert ref.decompress(ours, n) == data, (name, n)
            r = len(ours) / len(ref.compress(data, level=3))
            bar = max(1.05, recorded[name][str(n)]["size_ratio"] * 1.02)
            assert r <= bar, (name, n, r, bar)
            if r > 1.05:
                over[(name, n)] = round(r, 3)
    assert len(over) <= 6, over  # (the table's own count of cells above 1.05)


def test_device_entry_one_frame_with_checksum(ref):
    """The device-resident entry in the one-frame layout (on request there; its default is independent frames): with
    ZL_c_checksumFlag the frame's XXH64 is computed on the host over pieces brought down from the device and appended on the
    device; the last chunk's reported size covers those four bytes, which is what the decoder's chunk hint expects.  Same bytes
    as the host entry writes for the same input."""
    import torch
    from zstdlite_amd.device import DeviceCodec
    data = GENERATORS["text"](700001, seed=21) + GENERATORS["iid10"](300000, seed=22) + GENERATORS["rand"](100000, seed=23)
    n = len(data)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    codec = DeviceCodec(include_checksum=True, frame_layout="one_frame")
    dst, total, sizes = codec.compress(src)
    comp = bytes(dst[:total].cpu().numpy())
    assert int(sizes.sum()) == total
    assert comp == z.zstd_compress(data, include_checksum=True)
    assert ref.L.ZSTD_findFrameCompressedSize(comp, len(comp)) == len(comp)
    assert ref.decompress(comp, n) == data
    back = torch.zeros_like(src)
    tbl = codec.frame_table([total], [n])
    for hint in (sizes, None):
        back.zero_()
        got, _ = codec.decompress(dst, tbl, back, chunk_sizes=hint)
        assert got == n and torch.equal(back, src)
    bad = dst.clone()
    bad[total - 2] ^= 0x21
    with pytest.raises(z.ZstdError, match="doesn't match checksum"):
        codec.decompress(bad, tbl, back, chunk_sizes=sizes)
    codec.close()


def test_batch_entries_report_per_object_results_at_scale():
    """zl_compress_batch / zl_decompress_batch with tens of thousands of objects: the per-object loops of the call's host side
    run on several host threads (zl_host.h zl_parallel_for) and one launch group takes the whole batch (scratch strides sized
    to the largest object).  Every object's result stands at its own index: sizes, an empty object, a destination that is too
    small (zstd.h:1219 dstSize_tooSmall for that object only, the others unaffected), a source that is not a frame -- and the
    same bytes come out as object by object through zl_compress2."""
    import ctypes as C
    from zstdlite_amd import _lib
    rng = np.random.default_rng(5)
    L = _lib.lib()
    n = 40000
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 9)), dtype=np.uint8)) for _ in range(200)]
    items = [b" ".join(words[int(j)] for j in rng.integers(0, 200, int(rng.integers(0, 120)))) for _ in range(n)]
    items[17] = b""
    items[4242] = b"some words that repeat, some words that repeat"
    items[n - 1] = bytes(3000)
    sizes = np.array([len(x) for x in items], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    blob = np.frombuffer(b"".join(items), dtype=np.uint8).copy()
    caps = np.array([L.zl_compressBound(int(s)) for s in sizes], dtype=np.uint64)
    coff = np.concatenate([[0], np.cumsum(caps)]).astype(np.uint64)
    comp = np.zeros(int(coff[-1]), dtype=np.uint8)
    p_src = (blob.ctypes.data + off[:-1]).astype(np.uint64)
    p_comp = (comp.ctypes.data + coff[:-1]).astype(np.uint64)
    caps_c = caps.copy()
    small = [5, 20001, n - 2]
    for i in small:
        caps_c[i] = 3  # too small for any frame of a non-empty object
    assert all(sizes[i] > 0 for i in small)
    csize = np.zeros(n, dtype=np.uint64)
    vp, sp = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
    cctx, dctx = L.zl_createCCtx(), L.zl_createDCtx()
    try:
        r = L.zl_compress_batch(cctx, n, p_comp.ctypes.data_as(vp), caps_c.ctypes.data_as(sp), p_src.ctypes.data_as(vp), sizes.ctypes.data_as(sp), csize.ctypes.data_as(sp))
        assert not _lib.is_error(r)
        for i in range(n):
            if i in small:
                assert _lib.is_error(int(csize[i])) and "too small" in _lib.error_name(int(csize[i])).lower(), i
            else:
                assert not _lib.is_error(int(csize[i])) and 0 < csize[i] <= caps[i], i
        for i in (0, 17, 4242, 20000, 20002, n - 1):  # the same bytes as the one-object entry writes
            one = z.zstd_compress(items[i], frame_layout="frames")
            assert bytes(comp[int(coff[i]):int(coff[i]) + int(csize[i])]) == one, i
        # decompress: the objects whose compression failed are handed garbage instead of a frame
        dsizes = csize.copy()
        for i in small:
            dsizes[i] = 3
        back = np.zeros(len(blob) + 16, dtype=np.uint8)
        p_back = (back.ctypes.data + off[:-1]).astype(np.uint64)
        dcap = sizes.copy()
        dcap[4242] = max(int(sizes[4242]) - 1, 0)  # one destination too small
        bsize = np.zeros(n, dtype=np.uint64)
        r = L.zl_decompress_batch(dctx, n, p_back.ctypes.data_as(vp), dcap.ctypes.data_as(sp), p_comp.ctypes.data_as(vp), dsizes.ctypes.data_as(sp), bsize.ctypes.data_as(sp))
        assert not _lib.is_error(r)
        for i in range(n):
            if i in small or i == 4242:
                assert _lib.is_error(int(bsize[i])), i
            else:
                assert bsize[i] == sizes[i], i
                assert bytes(back[int(off[i]):int(off[i + 1])]) == items[i], i
    finally:
        L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)


def test_output_does_not_depend_on_scheduling_when_a_chunk_oversamples(ref):
    """The far search lists a chunk's sampled positions (content-defined: the top bits of the hash of 8 bytes), at most
    ZL_FAR_LIST_CAP of them.  A long run samples all of its positions or none, so a chunk of runs lists far more than the cap
    -- and WHICH ones were kept used to depend on the order in which the waves of zl_far_build_kernel reached an atomic counter:
    the same input compressed to different (valid) bytes from call to call (found by scripts/fuzz_encode.py at level 9, one
    frame).  Now such a chunk is listed again in position order.  Runs of 64 KiB of 24 byte values (every chunk holds two; at
    either sampling rate several of them are sampled) plus some text: ten calls, one output, at both operating points."""
    import warnings
    data = b"".join(bytes([b]) * 65536 for b in range(24)) + GENERATORS["text"](100000, seed=8)
    for level in (3, 5):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs = {z.zstd_compress(data, level=level, frame_layout="one_frame") for _ in range(10)}
        assert len(outs) == 1, (level, len(outs))
        assert ref.decompress(next(iter(outs)), len(data)) == data


@pytest.mark.parametrize("layout", ["one_frame", "frames"])
def test_a_seek_table_with_a_damaged_content_field_is_not_believed(layout, ref):
    """The seek table is a skippable frame: damage inside it must not change what the data in front of it decodes to
    (include/zl_hip.h: checked against the data, never trusted).  Until round 3 a content field that grew by a byte -- still the
    same number of chunks, so the table passed its own checks -- made zl_decompressDCtx fail with dstSize_tooSmall on a buffer the
    reference reads (scripts/fuzz_seek.py); every failure on the table's word now sends the call to the header walk."""
    import struct
    data = GENERATORS["text"](700000, seed=12)
    comp = bytearray(z.zstd_compress(data, frame_layout=layout, seek_table=True))
    plain = z.zstd_compress(data, frame_layout=layout)
    assert len(comp) > len(plain)
    if layout == "one_frame":   # private trailer: ... u64 content | u32 nchunks | u32 flags | u32 tag
        at = len(comp) - 20
        (content,) = struct.unpack_from("<Q", comp, at)
        assert content == len(data)
        struct.pack_into("<Q", comp, at, content + 1)
    else:                       # seekable format: the last frame's decompressed size sits 8 + 5 bytes in front of the end... entry {c, d}
        at = len(comp) - 9 - 4
        (dsz,) = struct.unpack_from("<I", comp, at)
        assert dsz == len(data) - (len(data) // 131072) * 131072
        struct.pack_into("<I", comp, at, dsz + 1)
    assert ref.decompress(bytes(comp), len(data)) == data
    assert z.zstd_decompress(bytes(comp)) == data


def test_deep_levels_degrade_to_the_table_parse_when_their_scratch_cannot_be_had(ref):
    """a deep-level call whose scratch does not fit (8 bytes per input byte, a section's key arrays, the sort's storage)
    or whose sort refuses used to FAIL with a memory error.  Now the entry makes the call again on the table parse at its deepest
    setting: valid frames, a little larger.  ZL_c_faultInjection bit 0 makes every deep allocation "fail"."""
    import torch
    from zstdlite_amd import _lib, stream
    from zstdlite_amd.device import DeviceCodec, gen_mixed
    L = _lib.lib()
    text = GENERATORS["text"](3 << 20, seed=5)
    denied, deep = [], []
    for level in (3, 9, 15, 19):
        cctx = z.zstd_cctx(level=level)
        assert L.zl_CCtx_setParameter(cctx._ptr, 9098, 1) == 0
        comp = z.zstd_compress(text, cctx=cctx)
        assert ref.decompress(comp, len(text)) == text and z.zstd_decompress(comp) == text
        denied.append(len(comp))
        deep.append(len(z.zstd_compress(text, level=level)))
    # (level 3 on this synthetic text is the one place where the tables write a little LESS than three candidates of the deep finder: DESIGN 3.3)
    assert denied[1] > deep[1] and denied[2] > deep[2] and denied[3] > deep[3] and denied[0] < 1.05 * deep[0], (denied, deep)
    # a pledged stream of several jobs with a content checksum: every job is made twice (the deep attempt, then the table parse), the
    # running XXH64 must cover the bytes once
    n = (150 << 20) + 5
    data = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
    cctx = z.zstd_cctx(level=9, include_checksum=True)
    assert L.zl_CCtx_setParameter(cctx._ptr, 9098, 1) == 0
    comp = stream.Compressor(cctx._ptr, pledged=n)
    blob = b"".join([comp.feed(data[off:off + 40000003]) for off in range(0, n, 40000003)] + [comp.finish()])
    assert ref.L.ZSTD_findFrameCompressedSize(blob, len(blob)) == len(blob)
    assert ref.decompress(blob, n) == data  # (the reference verifies the checksum)
    # the device-resident entry
    codec = DeviceCodec(level=9, frame_layout="one_frame")
    assert L.zl_CCtx_setParameter(codec.cctx, 9098, 1) == 0
    src = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    dst, total, _ = codec.compress(src)
    assert ref.decompress(bytes(dst[:total].cpu().numpy()), len(text)) == text


def test_small_objects_on_a_dictionary_at_the_edges_of_the_wave_kernels(ref, oracle):
    """Round 6 (config C4): a batch of small objects with a dictionary is encoded by zl_entropy_small_kernel (a wave per object: treeless
    literals and Repeat_Mode sequences on the dictionary's tables) and decoded by zl2_small_kernel (four frames per wave), with the
    general kernels behind them for whatever those cannot take.  Objects around every limit of the two -- 1023 / 1024 literals, 127 / 128
    sequences, 1008 / 1009 bytes of content, 16 KiB, bytes the dictionary's Huffman table cannot code, runs, nothing at all -- must come out as
    frames the REFERENCE decodes with the dictionary, the oracle too, and this library's batch decoder; a batch is deterministic and an
    object's bytes do not depend on its neighbours (the one-object entry writes the same); damaged frames get the oracle's verdict."""
    import pickle
    rng = np.random.default_rng(17)
    names = [f"name_{i:02d}_{'x' * (i % 7)}" for i in range(50)]

    def obj(k):
        perm = rng.permutation(50)[:k]
        return pickle.dumps({names[j]: int(rng.integers(0, 1000)) for j in perm}, protocol=4)
    train = [obj(int(rng.integers(10, 40))) for _ in range(1500)]
    d = ref.train_dict(train, 5000)
    base = b"".join(train[:60])
    items = [obj(k) for k in (1, 2, 5, 10, 25, 39, 50)] * 3
    items += [base[:k] for k in (1, 2, 3, 5, 6, 7, 63, 64, 255, 256, 257, 1007, 1008, 1009, 1023, 1024, 1025, 2047, 2048, 4000, 16383, 16384, 16385, 40000)]
    items += [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in (1, 6, 100, 1023, 1024, 3000)]            # no table codes these: raw
    items += [bytes(rng.integers(97, 123, int(k), dtype=np.uint8)) for k in (50, 700, 1023, 1024, 1500)]             # literals only, all codable
    items += [b"ab" * k for k in (3, 40, 500, 600)] + [b"x" * k for k in (1, 7, 64, 1000, 5000)] + [b"", bytes(300)]
    items += [b"".join(train[i][:9] + bytes([i & 255]) for i in range(k)) for k in (100, 126, 127, 128, 129, 200)]   # many short sequences
    cctx = z.zstd_cctx(dict=d)
    frames = z.zstd_compress_batch(items, cctx=cctx)
    assert frames == z.zstd_compress_batch(items, cctx=cctx)
    assert frames[:12] == [z.zstd_compress(x, cctx=cctx) for x in items[:12]]
    for x, f in zip(items, frames):
        assert ref.decompress(f, max(len(x), 1), dict=d) == x, len(x)
        assert oracle.decompress(f, cap=max(len(x), 1), dict=d) == x, len(x)
    assert z.zstd_decompress_batch(frames, dict=d) == items
    # what the REFERENCE writes for the same objects (its own choice of tables and streams) through the batch decoder
    theirs = [ref.compress(x, level=3, dict=d) for x in items]
    assert z.zstd_decompress_batch(theirs, dict=d) == items
    # damaged frames (one to three flipped bits), ours and the reference's, through the batch entry: accepted exactly when the oracle accepts
    # them, and then the same bytes (zl2_small_kernel only ever accepts: what fails any of its checks is decoded -- or named -- by the lanes)
    import ctypes as C
    from zstdlite_amd import _lib
    from zl_testlib import is_error
    damaged = []
    for f, x in list(zip(frames, items)) + list(zip(theirs, items)):
        if len(x) > 3000 or len(f) < 12:
            continue
        for _ in range(12):
            b = bytearray(f)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
            damaged.append((bytes(b), len(x)))
    m = len(damaged)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for _, n in damaged]
    srcs = [np.frombuffer(f, dtype=np.uint8) for f, _ in damaged]
    res = (C.c_size_t * m)()
    dctx = z.zstd_dctx(dict=d)
    r = _lib.lib().zl_decompress_batch(dctx._ptr, m, (C.c_void_p * m)(*[x.ctypes.data for x in outs]), (C.c_size_t * m)(*[n for _, n in damaged]),
                                       (C.c_void_p * m)(*[x.ctypes.data for x in srcs]), (C.c_size_t * m)(*[len(f) for f, _ in damaged]), res)
    assert not is_error(r)
    accepted = 0
    for i, (f, n) in enumerate(damaged):
        code, out = oracle.decompress_code(f, n, dict=d)
        if is_error(res[i]):
            assert is_error(code), (i, _lib.error_name(res[i]), code)
        else:
            assert not is_error(code), (i, res[i], code)
            assert res[i] == code and outs[i][:res[i]].tobytes() == out, i
            accepted += 1
    assert 0 < accepted < m
