"""Pins the oracle (oracle/zl_oracle.c, this repo's CPU restatement of the reference decode
path) against (1) the reference's own binary known-answer fixture, (2) golden frames made by
the reference libzstd and committed under tests/golden/, (3) the live reference build when
oracle/_ref is present.  CPU only."""
import numpy as np
import pytest

from zl_testlib import GENERATORS, GOLDEN, Oracle, is_error, sha256
import os

# README.md:211-223 of the reference: plaintext of man/figures/data.json.zst
DATA_JSON = (b'{\n  "mpg": [21],\n  "cyl": [6],\n  "disp": [160],\n  "hp": [110],\n  "drat": [3.9],\n'
             b'  "wt": [2.62],\n  "qsec": [16.46],\n  "vs": [0],\n  "am": [1],\n  "gear": [4],\n  "carb": [4]\n}\n')


def test_known_answer_fixture(oracle):
    z = open(os.path.join(GOLDEN, "data.json.zst"), "rb").read()
    assert len(z) == 126
    assert oracle.L.zlo_find_frame_compressed_size(z, len(z)) == 126
    assert oracle.L.zlo_get_frame_content_size(z, len(z)) == 172  # FCS 172, checksum flag set
    assert (z[4] >> 2) & 1 == 1
    assert oracle.decompress(z) == DATA_JSON


def test_golden_frames(oracle, golden):
    cases, d = golden
    assert len(cases) >= 150
    for c in cases:
        out = oracle.decompress(c["comp"], dict=d if c["gen"] == "dict" else None)
        assert out == c["plain"], c["key"]
        assert oracle.L.zlo_find_frame_compressed_size(c["comp"], len(c["comp"])) == c["csize"]
        assert oracle.L.zlo_get_frame_content_size(c["comp"], len(c["comp"])) == c["n"]


def test_golden_large_frames(oracle, golden_large):
    """Reference-made single frames of more than eight blocks (tests/golden/large_frames.npz): the oracle
    is pinned against them too -- they are what the large-frame pipeline is checked with on the GPU."""
    assert len(golden_large) >= 3
    for c in golden_large:
        assert oracle.decompress(c["comp"]) == c["plain"], c["key"]
        assert oracle.L.zlo_find_frame_compressed_size(c["comp"], len(c["comp"])) == c["csize"]
        assert oracle.L.zlo_get_frame_content_size(c["comp"], len(c["comp"])) == c["n"] > 8 * 131072


def test_pinned_facts(oracle, golden):
    # BASELINE.md section 2
    assert oracle.L.zlo_compress_bound(1 << 30) == 1077936128
    assert oracle.L.zlo_compress_bound(65536) == 65824
    _, d = golden
    assert oracle.L.zlo_dict_id(d, len(d)) != 0


def test_xxh64_known_answers(oracle):
    # public xxHash64 test vectors
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"a") == 0xD24EC4F1A98C6E5B
    assert oracle.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert oracle.xxh64(b"Nobody inspects the spammish repetition") == 0xFBCEA83C8A378BF1


def test_checksum_and_errors(oracle, golden):
    cases, _ = golden
    c = next(c for c in cases if c["checksum"] and c["n"] == 1000 and c["gen"] == "text")
    bad = bytearray(c["comp"])
    bad[-1] ^= 0xFF
    code, _ = oracle.decompress_code(bytes(bad), c["n"])
    assert is_error(code) and "doesn't match checksum" in oracle.error_name(code)  # test-checksums.R:24-27
    code, out = oracle.decompress_code(bytes(bad), c["n"], ignore_checksum=True)
    assert not is_error(code) and out == c["plain"]  # test-checksums.R:29-32
    code, _ = oracle.decompress_code(c["comp"][:-5], c["n"])
    assert is_error(code)
    code, _ = oracle.decompress_code(b"\x00\x01\x02\x03\x04\x05\x06\x07", 10)
    assert oracle.error_name(code) == "Unknown frame descriptor"
    code, _ = oracle.decompress_code(c["comp"], c["n"] - 1)
    assert oracle.error_name(code) == "Destination buffer is too small"


def test_multiframe_and_skippable(oracle, golden):
    cases, _ = golden
    a = next(c for c in cases if c["gen"] == "text" and c["n"] == 4096 and c["level"] == 3)
    b = next(c for c in cases if c["gen"] == "iid10" and c["n"] == 1000 and c["level"] == 3)
    skip = b"\x53\x2a\x4d\x18" + (5).to_bytes(4, "little") + b"hello"
    blob = a["comp"] + skip + b["comp"]
    assert oracle.decompress(blob, cap=a["n"] + b["n"]) == a["plain"] + b["plain"]


@pytest.mark.parametrize("gen", ["text", "iid10", "int20", "rand"])
def test_live_reference_differential(oracle, ref, gen):
    """Wider differential run against the live reference build (skipped on boxes without it)."""
    assert ref.version() == "1.5.6"
    rng = np.random.default_rng(7)
    for n in [3, 50, 700, 5000, 70000, 140000, 400000]:
        data = GENERATORS[gen](n, seed=int(rng.integers(1 << 30)))
        for level in (3, 6, 12, -1):
            comp = ref.compress(data, level=level, checksum=bool(level == 6))
            assert oracle.decompress(comp) == data
            assert oracle.L.zlo_find_frame_compressed_size(comp, len(comp)) == ref.L.ZSTD_findFrameCompressedSize(comp, len(comp))
    # Corrupted streams.  The reference's Huffman fast path (zstd.c:38890-38940) only checks
    # that each stream produced its segment, not that its bits were consumed exactly, so it
    # accepts some damaged frames that a format-strict decoder rejects.  The oracle (and the
    # HIP decoder) are format-strict; what must hold: whatever the oracle accepts the reference
    # accepts with identical bytes, and whatever the reference rejects the oracle rejects.
    data = GENERATORS[gen](20000)
    comp = bytearray(ref.compress(data, level=3))
    for k in range(60):
        pos = int(rng.integers(4, len(comp)))
        mutated = bytearray(comp)
        mutated[pos] ^= 1 << int(rng.integers(8))
        rc, rout = ref.decompress_code(bytes(mutated), len(data))
        oc, oout = oracle.decompress_code(bytes(mutated), len(data))
        if not is_error(oc):
            assert not ref.L.ZSTD_isError(rc) and rout == oout, (gen, k, pos)
        if ref.L.ZSTD_isError(rc):
            assert is_error(oc), (gen, k, pos, ref.error_name(rc))


def test_c1_shape_mtcars_head_serialization(oracle, ref):
    """BASELINE config C1 (zstd_serialize(head(mtcars)) round trip via the bundled CPU libzstd) without R: the 1000-byte
    native-binary serialization built by hand (tests/golden/make_c1.py) goes through the reference's call sequence
    (src/serialize.c:73-91,171-197).  README.md:96-105 pins the shape: frame header 28 b5 2f fd 60 e8 02 -- content size
    1000 in a single-segment frame -- and 482 bytes in all (our buffer differs from R's own in a few bytes the README
    does not show, so the size is checked to a few per cent, the header exactly)."""
    import sys
    sys.path.insert(0, os.path.join(GOLDEN))
    import make_c1
    buf = open(os.path.join(GOLDEN, "c1_mtcars_head.bin"), "rb").read()
    assert len(buf) == 1000 and buf == make_c1.build()
    comp = ref.compress(buf, level=3)
    assert comp[:7] == make_c1.README_FIRST_100[:7]
    assert abs(len(comp) - 482) <= 24
    assert ref.L.ZSTD_getFrameContentSize(comp, len(comp)) == 1000 and ref.L.ZSTD_findFrameCompressedSize(comp, len(comp)) == len(comp)
    assert oracle.decompress(comp) == buf and ref.decompress(comp, 1000) == buf
