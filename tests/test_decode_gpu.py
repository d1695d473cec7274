"""Parity tests proper for the HIP decoder: every call goes through the C-ABI
(libzl_hip.so) and is compared bit-for-bit with the oracle / the committed golden frames
made by the reference libzstd.  Mirrors tests/testthat/test-compress-raw.R,
test-checksums.R of the reference (round-trip identity, checksum error text)."""
import os

import numpy as np
import pytest

import zstdlite_amd as z
from zstdlite_amd import _lib
from zl_testlib import GENERATORS, GOLDEN, have_ref, is_error

pytestmark = pytest.mark.gpu


def test_known_answer_fixture(oracle):
    zj = open(os.path.join(GOLDEN, "data.json.zst"), "rb").read()
    out = z.zstd_decompress(zj)
    assert out == oracle.decompress(zj)
    assert z.zstd_decompress(zj, type="string").startswith('{\n  "mpg": [21]')  # README.md:211


def test_golden_frames_bit_exact(golden, oracle):
    cases, d = golden
    dctx = z.zstd_dctx()
    dctx_dict = z.zstd_dctx(dict=d)
    for c in cases:
        ctx = dctx_dict if c["gen"] == "dict" else dctx
        assert z.zstd_decompress(c["comp"], dctx=ctx) == c["plain"], c["key"]


def test_batch_decode_matches(golden):
    cases, d = golden
    plain = [c for c in cases if c["gen"] != "dict"]
    outs = z.zstd_decompress_batch([c["comp"] for c in plain])
    for c, o in zip(plain, outs):
        assert o == c["plain"], c["key"]
    dc = [c for c in cases if c["gen"] == "dict"]
    outs = z.zstd_decompress_batch([c["comp"] for c in dc], dict=d)
    assert outs == [c["plain"] for c in dc]


def test_multiframe_skippable_concatenation(golden, oracle):
    cases, _ = golden
    picks = [c for c in cases if c["gen"] in ("text", "iid10") and c["level"] == 3 and c["n"] in (1000, 65536, 200000)]
    skip = b"\x5a\x2a\x4d\x18" + (7).to_bytes(4, "little") + b"skip me"
    blob = b"".join(c["comp"] + skip for c in picks)
    want = b"".join(c["plain"] for c in picks)
    assert z.zstd_decompress(blob) == want == oracle.decompress(blob, cap=len(want))


def test_checksum_error_text_and_ignore(golden):
    cases, _ = golden
    c = next(c for c in cases if c["checksum"] and c["gen"] == "text" and c["n"] == 65536)
    assert z.zstd_decompress(c["comp"]) == c["plain"]
    bad = bytearray(c["comp"])
    bad[-1] ^= 0x5A
    with pytest.raises(z.ZstdError, match="doesn't match checksum"):  # tests/testthat/test-checksums.R:22-27
        z.zstd_decompress(bytes(bad))
    assert z.zstd_decompress(bytes(bad), validate_checksum=False) == c["plain"]  # test-checksums.R:29-32
    assert z.zstd_decompress(bytes(bad), dctx=z.zstd_dctx(validate_checksum=False)) == c["plain"]


def test_malformed_inputs_report_errors(golden, oracle):
    cases, d = golden
    c = next(c for c in cases if c["gen"] == "text" and c["n"] == 65536 and c["level"] == 3)
    with pytest.raises(z.ZstdError, match="Unknown frame descriptor"):
        z.zstd_decompress(b"not a zstd frame at all")
    with pytest.raises(z.ZstdError, match="Src size is incorrect"):
        z.zstd_decompress(c["comp"][:-7])
    dc = next(c for c in cases if c["gen"] == "dict")
    with pytest.raises(z.ZstdError, match="Dictionary mismatch"):  # zstd.c:41318
        z.zstd_decompress(dc["comp"])
    # bit flips: same accept/reject decision and bytes as the (format-strict) oracle
    L = _lib.lib()
    rng = np.random.default_rng(3)
    dctx = z.zstd_dctx()
    import ctypes as C
    for k in range(40):
        m = bytearray(c["comp"])
        m[int(rng.integers(6, len(m)))] ^= 1 << int(rng.integers(8))
        m = bytes(m)
        oc, oout = oracle.decompress_code(m, c["n"])
        dst = np.empty(c["n"], dtype=np.uint8)
        r = L.zl_decompressDCtx(dctx._ptr, dst.ctypes.data_as(C.c_void_p), c["n"], m, len(m))
        assert is_error(r) == is_error(oc), (k, _lib.error_name(r), oracle.error_name(oc))
        if not is_error(r):
            assert dst[:r].tobytes() == oout


@pytest.mark.parametrize("gen", ["text", "iid10", "int20", "rand"])
def test_live_reference_frames(gen, ref):
    """Frames freshly made by the reference libzstd (oracle/_ref, which travels to the GPU box):
    ragged sizes, several levels, multi-block frames, checksums."""
    rng = np.random.default_rng(5)
    for n in [0, 1, 2, 3, 11, 255, 256, 257, 65535, 65536, 65537, 131071, 131072, 131073, 500000, 1 << 20]:
        data = GENERATORS[gen](n, seed=int(rng.integers(1 << 30)))
        for level in ((3, 1, 19) if n <= 131073 else (3,)):
            comp = ref.compress(data, level=level, checksum=(level == 1))
            assert z.zstd_decompress(comp) == data, (gen, n, level)


def test_c2_shape_many_64k_frames(ref):
    """BASELINE config C2 at reduced count: independent 64 KiB frames made by the reference at
    level 3, one third each text / iid10 / rand, decoded in one launch."""
    frames, plains = [], []
    for i in range(96):
        gen = ("text", "iid10", "rand")[i % 3]
        data = GENERATORS[gen](65536, seed=1000 + i)
        frames.append(ref.compress(data, level=3))
        plains.append(data)
    assert z.zstd_decompress(b"".join(frames)) == b"".join(plains)
    assert z.zstd_decompress_batch(frames) == plains


def _ref_stream_frame(ref, pieces, level=3, pledged=True):
    """One frame, one block per piece: the reference's streaming entry with ZSTD_e_flush between
    pieces (what src/raw-file-out.c does) -- gives multi-block frames whose later blocks reuse the
    entropy tables (repeat / treeless modes) and copy from earlier blocks."""
    import ctypes as C

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    L = ref.L
    L.ZSTD_compressStream2.restype = C.c_size_t
    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(Buf), C.POINTER(Buf), C.c_int]
    L.ZSTD_CCtx_setPledgedSrcSize.restype = C.c_size_t
    L.ZSTD_CCtx_setPledgedSrcSize.argtypes = [C.c_void_p, C.c_ulonglong]
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, level)
    total = sum(len(p) for p in pieces)
    if pledged:
        L.ZSTD_CCtx_setPledgedSrcSize(cctx, total)
    cap = L.ZSTD_compressBound(total) + 64 * len(pieces)
    dst = C.create_string_buffer(cap)
    ob = Buf(C.cast(dst, C.c_void_p), cap, 0)
    for i, piece in enumerate(pieces):
        src = C.create_string_buffer(bytes(piece), len(piece))
        ib = Buf(C.cast(src, C.c_void_p), len(piece), 0)
        mode = 2 if i == len(pieces) - 1 else 1  # ZSTD_e_end / ZSTD_e_flush
        while True:
            r = L.ZSTD_compressStream2(cctx, C.byref(ob), C.byref(ib), mode)
            assert not L.ZSTD_isError(r)
            if r == 0 and ib.pos == ib.size:
                break
    L.ZSTD_freeCCtx(cctx)
    return dst.raw[:ob.pos]


def test_multiblock_frames_with_repeated_tables(ref, oracle):
    """Frames of <= 128 KiB made of several blocks go through the lane-parallel pipeline with
    table reuse and cross-block matches; larger ones and those without a recorded size through the
    general kernel.  Both must equal the plaintext (and the oracle)."""
    from zl_testlib import GENERATORS
    text = GENERATORS["text"](120000, seed=9)
    iid = GENERATORS["iid10"](60000, seed=9)
    cases = [
        [text[i:i + 30000] for i in range(0, 120000, 30000)],             # 4 blocks, same statistics -> repeat modes
        [text[:20000], text[:20000], text[5000:25000]],                    # later blocks are copies of earlier ones
        [iid[:10000], text[:10000], iid[10000:30000], text[10000:30000]],  # alternating statistics
        [text[i:i + 9000] for i in range(0, 90000, 9000)],                 # 10 blocks: more than the pipeline takes -> handed back
        [text[:70000], iid, text[70000:]],                                 # > 128 KiB -> general kernel
    ]
    for pieces in cases:
        plain = b"".join(pieces)
        for pledged in (True, False):
            frame = _ref_stream_frame(ref, pieces, pledged=pledged)
            assert oracle.decompress(frame, cap=len(plain)) == plain
            if pledged:
                assert z.zstd_decompress(frame) == plain
            else:  # no recorded size: the C-ABI decodes into whatever capacity the caller gives
                import ctypes as C
                L = _lib.lib()
                dctx = z.zstd_dctx()
                dst = np.empty(len(plain) + 100, dtype=np.uint8)
                r = L.zl_decompressDCtx(dctx._ptr, dst.ctypes.data_as(C.c_void_p), dst.size, frame, len(frame))
                assert not is_error(r), _lib.error_name(r)
                assert dst[:r].tobytes() == plain
    # two such frames concatenated, second one without size
    a = _ref_stream_frame(ref, cases[0], pledged=True)
    b = _ref_stream_frame(ref, cases[2], pledged=False)
    import ctypes as C
    L = _lib.lib()
    want = b"".join(cases[0]) + b"".join(cases[2])
    dst = np.empty(len(want), dtype=np.uint8)
    dctx = z.zstd_dctx()  # keep the context alive for the call
    r = L.zl_decompressDCtx(dctx._ptr, dst.ctypes.data_as(C.c_void_p), dst.size, a + b, len(a) + len(b))
    assert not is_error(r), _lib.error_name(r)
    assert dst[:r].tobytes() == want


def test_damaged_frames_agree_with_the_oracle(oracle):
    """Frames of the HIP encoder (8 blocks sharing tables: symbolic repcodes, leader rounds) and of the
    reference library with 1-3 flipped bits, decoded 1000+ at a time: the GPU must accept exactly the
    frames the oracle accepts and produce the same bytes (scripts/fuzz_decode.py is the long version)."""
    import ctypes as C
    from zstdlite_amd import _lib
    from zl_testlib import GENERATORS, Ref, have_ref, is_error
    rng = np.random.default_rng(11)
    base = []
    for gen in ("text", "iid10", "int20"):
        for n in (300, 20000, 131072):
            base.append((z.zstd_compress(GENERATORS[gen](n, seed=n)), n))
    if have_ref():
        ref = Ref()
        for gen in ("text", "iid10"):
            for n in (65536, 300000):
                base.append((ref.compress(GENERATORS[gen](n, seed=n + 1), level=3), n))
    frames = []
    for f, n in base:
        for _ in range(80):
            b = bytearray(f)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
            frames.append((bytes(b), n))
    m = len(frames)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for _, n in frames]
    srcs = [np.frombuffer(f, dtype=np.uint8) for f, _ in frames]
    res = (C.c_size_t * m)()
    dctx = z.zstd_dctx()
    r = _lib.lib().zl_decompress_batch(dctx._ptr, m, (C.c_void_p * m)(*[x.ctypes.data for x in outs]), (C.c_size_t * m)(*[n for _, n in frames]),
                                       (C.c_void_p * m)(*[x.ctypes.data for x in srcs]), (C.c_size_t * m)(*[len(f) for f, _ in frames]), res)
    assert not is_error(r)
    accepted = 0
    for i, (f, n) in enumerate(frames):
        code, out = oracle.decompress_code(f, n)
        if is_error(res[i]):
            assert is_error(code), (i, _lib.error_name(res[i]), code)
        else:
            accepted += 1
            assert not is_error(code) and res[i] == code and outs[i][: res[i]].tobytes() == out, i
    assert 0 < accepted < m


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (reference libzstd) to make the frames")
def test_damaged_single_block_frames_in_a_batch_that_is_cut_in_two(oracle):
    """A batch of 1024+ single-block frames is dealt out in two parts that run side by side (zl2_launch: the cheap half's
    entropy and exec kernels next to the expensive half's chains).  Reference-made frames of one block each, most of them
    damaged by 1-3 flipped bits: the GPU must accept exactly the frames the oracle accepts, with the same bytes."""
    import ctypes as C
    from zstdlite_amd import _lib
    from zl_testlib import GENERATORS, Ref, is_error
    ref = Ref()
    rng = np.random.default_rng(23)
    base = []
    for gen in ("text", "iid10", "rand"):
        for n in (3000, 30000, 65536):
            base.append((ref.compress(GENERATORS[gen](n, seed=n + 3), level=3), n))
    frames = []
    for f, n in base:
        for k in range(130):
            b = bytearray(f)
            if k % 5:  # every fifth frame stays whole
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(4, len(b)))] ^= 1 << int(rng.integers(0, 8))
            frames.append((bytes(b), n))
    m = len(frames)
    assert m >= 1024
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for _, n in frames]
    srcs = [np.frombuffer(f, dtype=np.uint8) for f, _ in frames]
    res = (C.c_size_t * m)()
    dctx = z.zstd_dctx()
    r = _lib.lib().zl_decompress_batch(dctx._ptr, m, (C.c_void_p * m)(*[x.ctypes.data for x in outs]), (C.c_size_t * m)(*[n for _, n in frames]),
                                       (C.c_void_p * m)(*[x.ctypes.data for x in srcs]), (C.c_size_t * m)(*[len(f) for f, _ in frames]), res)
    assert not is_error(r)
    accepted = 0
    for i, (f, n) in enumerate(frames):
        code, out = oracle.decompress_code(f, n)
        if is_error(res[i]):
            assert is_error(code), (i, _lib.error_name(res[i]), code)
        else:
            accepted += 1
            assert not is_error(code) and res[i] == code and outs[i][: res[i]].tobytes() == out, i
    assert accepted >= m // 5


@pytest.mark.gpu
def test_dictionary_tables_survive_frames_that_bring_their_own(ref):
    """The general kernel keeps the dictionary's entropy tables in LDS from one frame to the next and copies again only what a
    frame in between has replaced (zl_decode.hip: DecShared::dict_live).  A batch that mixes, in one launch and in an order the
    workgroups pick, frames that ride on the dictionary's tables (small objects: Repeat_Mode, treeless literals), frames that
    describe tables of their own (larger texts), frames with a content checksum (whose check stages the output in the
    Huffman table's LDS) and raw / RLE frames: every frame decodes to its input with the dictionary, made by the reference
    library (zstd.c:42053 ZSTD_loadDEntropy, :43666 repeat modes)."""
    import pickle
    rng = np.random.default_rng(11)
    names = [f"field_{i:02d}_{'y' * (i % 5)}" for i in range(40)]

    def small():
        perm = rng.permutation(40)[: int(rng.integers(8, 30))]
        return pickle.dumps({names[j]: int(rng.integers(0, 500)) for j in perm}, protocol=4)
    smalls = [small() for _ in range(1200)]
    d = ref.train_dict(smalls[:600], 4000)
    items = []
    for i in range(900):
        k = i % 9
        if k in (0, 1, 2, 3, 4):
            items.append((smalls[i], False))
        elif k == 5:
            items.append((GENERATORS["text"](int(rng.integers(20000, 60000)), seed=i), False))  # its own tables
        elif k == 6:
            items.append((smalls[i] * 2, True))                                                  # with a checksum
        elif k == 7:
            items.append((bytes(rng.integers(0, 256, int(rng.integers(50, 400)), dtype=np.uint8)), False))  # raw block
        else:
            items.append((GENERATORS["iid10"](int(rng.integers(3000, 9000)), seed=i), True))     # own tables + checksum
    frames = [ref.compress(x, level=3, checksum=c, dict=d) for x, c in items]
    modes = {f[5 + 4 + (1 if len(x) < 256 else (2 if len(x) < 65792 else 4)) + 3] & 3 for (x, _), f in zip(items, frames)}
    assert 3 in modes and 2 in modes, modes  # treeless literals and literals with their own tree are both in the batch
    for rep in range(3):  # (the order in which workgroups take the frames differs from launch to launch)
        back = z.zstd_decompress_batch(frames, dict=d)
        assert back == [x for x, _ in items]
    # a damaged checksum is still caught in the middle of such a batch
    i = next(k for k, (_, c) in enumerate(items) if c)
    bad = list(frames)
    bad[i] = frames[i][:-1] + bytes([frames[i][-1] ^ 0x5A])
    with pytest.raises(z.ZstdError):
        z.zstd_decompress_batch(bad, dict=d)


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (reference libzstd) to make the frames and the dictionary")
def test_dictionary_frames_on_the_lane_pipeline_agree_with_the_oracle(ref, oracle):
    """Frames of 2 KiB of content and more that start from a dictionary take the lane pipeline (zl_api.hip ZL_DICT_LANE_MIN):
    its sequence and literal lanes inherit the dictionary's tables from a synthetic block (zl2_args::ext_shared), chains start
    from a symbolic history that the exec kernel fills with the dictionary's repeat offsets, matches reach into the
    dictionary's content in front of position 0.  Reference-made frames (zstd.c:42053 ZSTD_loadDEntropy, :43640 repeat modes,
    :44030 ZSTD_execSequence with a dictionary) of one to five blocks, with and without checksum, from both a trained and a
    raw-content dictionary -- intact: bit-exact; with flipped bits: accepted exactly when the oracle accepts them, and then the
    same bytes."""
    import ctypes as C
    rng = np.random.default_rng(23)
    text = GENERATORS["text"](3 << 20, seed=31)
    trained = ref.train_dict([text[i * 4000:(i + 1) * 4000] for i in range(300)], 16000)
    raw = text[100000:130000]  # no magic: content only
    for d in (trained, raw):
        items = []
        for n in (2048, 3000, 9000, 40000, 131072, 131073, 300000, 600000):
            x = text[(1 << 20) + 7 * n:(1 << 20) + 8 * n]
            if n == 9000:
                x = x[:4000] + text[100100:104100] + x[8000:]  # long matches into the dictionary's content
            items.append(x)
        items.append(GENERATORS["iid10"](50000, seed=2))
        frames = [ref.compress(x, level=3, checksum=bool(i & 1), dict=d) for i, x in enumerate(items)]
        assert z.zstd_decompress_batch(frames, dict=d) == items
        assert [z.zstd_decompress(f, dict=d) for f in frames[:4]] == items[:4]
        # without the dictionary: whatever the reference says (a frame that names a dictionary id is refused, zstd.c:41318)
        for f, x in zip(frames[:3], items[:3]):
            code, _ = ref.decompress_code(f, len(x))
            if is_error(code):
                with pytest.raises(z.ZstdError):
                    z.zstd_decompress(f)
        damaged = []
        for f, x in zip(frames, items):
            for _ in range(40):
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
                accepted += 1
                assert not is_error(code) and res[i] == code and outs[i][: res[i]].tobytes() == out, i
        assert 0 < accepted < m


@pytest.mark.gpu
@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (reference libzstd) to make the frames")
def test_batch_entry_takes_frames_without_a_recorded_content_size(ref, oracle):
    """zl_decompress_batch decodes a frame that records no content size into the room its caller gave it, as
    ZSTD_decompressDCtx does (zstd.c:41554) -- round 2 refused such items ("Unsupported frame parameter").  Found by
    scripts/fuzz_dict.py: clearing the Single_Segment bit of a small frame turns its one-byte size field into a window
    descriptor and leaves a frame every other decoder accepts.  Also frames the reference's streaming API writes without a
    pledged size, with and without a dictionary; a destination that is too small for such a frame is that item's error only."""
    import ctypes as C
    x = GENERATORS["text"](100, seed=3)
    f = bytearray(ref.compress(x, level=3))
    assert f[4] & 0x20 and (f[4] >> 6) == 0      # Single_Segment with a one-byte Frame_Content_Size
    f[4] &= ~0x20                                 # ... which now reads as Window_Descriptor: content size unknown
    f = bytes(f)
    code, out = oracle.decompress_code(f, 4096)
    assert not is_error(code) and out == x
    y = GENERATORS["text"](50000, seed=4)
    g = bytearray(ref.compress(y, level=3))       # 3-byte size field (flag 1 with Single_Segment clear is still a recorded size)
    frames = [f, bytes(g), f, f]
    caps = [4096, 50000, 100, 99]                 # the last one: too small
    m = len(frames)
    outs = [np.zeros(max(c, 1), dtype=np.uint8) for c in caps]
    srcs = [np.frombuffer(b, dtype=np.uint8) for b in frames]
    res = (C.c_size_t * m)()
    dctx = z.zstd_dctx()
    r = _lib.lib().zl_decompress_batch(dctx._ptr, m, (C.c_void_p * m)(*[o.ctypes.data for o in outs]), (C.c_size_t * m)(*caps),
                                       (C.c_void_p * m)(*[s.ctypes.data for s in srcs]), (C.c_size_t * m)(*[len(b) for b in frames]), res)
    assert not is_error(r)
    assert res[0] == 100 and outs[0][:100].tobytes() == x
    assert res[1] == 50000 and outs[1].tobytes() == y
    assert res[2] == 100 and outs[2][:100].tobytes() == x
    assert is_error(res[3])
    one = np.zeros(4096, dtype=np.uint8)          # the one-shot entry of the C-ABI always took it
    r1 = _lib.lib().zl_decompressDCtx(dctx._ptr, one.ctypes.data, 4096, f, len(f))
    assert r1 == 100 and one[:100].tobytes() == x
    with pytest.raises(z.ZstdError):              # (the R glue's mirror refuses it before any decoder runs, as src/raw-file.c does)
        z.zstd_decompress(f)


def test_the_frame_table_walked_in_parts_is_the_serial_walk(ref):
    """zl_decompressDCtx walks the frames of a large input (32 MiB and more) in PARTS side by side, every part but the first from the first
    frame magic behind its start, and takes a part's list only where the list in front of it ends exactly there (zl_api.hip: ZL_WALK_PARTS).
    Frame magics INSIDE frame data (random bytes are stored as they are), skippable frames, frames longer than a part and a cut-off
    input: the same bytes and the same verdicts as the reference."""
    import struct
    L = _lib.lib()
    rng = np.random.default_rng(77)
    magic = bytes([0x28, 0xB5, 0x2F, 0xFD])
    parts, want = [], []
    total = 0
    k = 0
    while total < (40 << 20):
        nbytes = int(rng.choice([300000, 1 << 20, 3 << 20, 9 << 20])) if k != 5 else (14 << 20)  # (one frame longer than a part of an eighth)
        x = bytearray(rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes())
        for at in range(1000, nbytes - 8, 65536):  # false frame starts all over the stored bytes
            x[at:at + 4] = magic
            x[at + 4] = int(rng.integers(0, 256))
        x = bytes(x)
        f = ref.compress(x, level=1) if k % 3 else z.zstd_compress(x, frame_layout="frames")
        parts.append(f); want.append(x); total += len(f)
        if k % 4 == 1:
            pay = bytes(rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8))
            parts.append(struct.pack("<II", 0x184D2A50 + k % 16, len(pay)) + pay); total += 8 + len(pay)
        k += 1
    buf, plain = b"".join(parts), b"".join(want)
    dctx = z.zstd_dctx()
    back = np.zeros(len(plain), dtype=np.uint8)
    r = L.zl_decompressDCtx(dctx._ptr, back.ctypes.data, len(plain), buf, len(buf))
    assert r == len(plain) and back.tobytes() == plain
    for cut in (len(buf) - 1, len(buf) // 2, len(buf) // 8 + 5, 33 << 20):
        rc, _ = ref.decompress_code(buf[:cut], len(plain))
        r = L.zl_decompressDCtx(dctx._ptr, back.ctypes.data, len(plain), buf[:cut], cut)
        assert _lib.is_error(r) == is_error(rc), cut
    # (and a buffer whose every part starts inside a frame that holds no magic at all: one frame of zeros behind a small one)
    zeros = bytes(600 << 20)
    small = ref.compress(b"abc" * 1000, level=3)
    f0 = ref.compress(zeros, level=3)
    assert len(f0) < (1 << 20)  # (too small for the parts: the serial walk)
    big_rand = rng.integers(0, 256, 34 << 20, dtype=np.uint8).tobytes()
    f1 = ref.compress(big_rand, level=1)  # ONE frame of 34 MiB stored: every part lies inside it
    buf2, plain2 = small + f1 + small, b"abc" * 1000 + big_rand + b"abc" * 1000
    back2 = np.zeros(len(plain2), dtype=np.uint8)
    r = L.zl_decompressDCtx(dctx._ptr, back2.ctypes.data, len(plain2), buf2, len(buf2))
    assert r == len(plain2) and back2.tobytes() == plain2
