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
