"""Streaming entries: the twins of ZSTD_compressStream2 / ZSTD_decompressStream and the
file / connection glue on top of them.  Interop is the bar, as in the reference's own tests
(tests/testthat/test-compress-raw.R:38-105, test-zstdfile.R): what the streaming compressor writes,
the REFERENCE's ZSTD_decompressStream and one-shot decoder read; what the reference's streaming
compressor writes (frames without a recorded size when nothing is pledged), the streaming decoder reads."""
import ctypes as C
import os
import pickle

import numpy as np
import pytest

import zstdlite_amd as z
from zstdlite_amd import _lib, stream
from zl_testlib import GENERATORS, have_ref

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not have_ref(), reason="needs the reference build (oracle/_ref)")


def ref_stream_compress(ref, data, level=3, pledge=False, piece=100000, checksum=False):
    """ZSTD_compressStream2 of the reference library, fed in pieces (src/raw-file-out.c:83-123)."""
    L = ref.L
    L.ZSTD_compressStream2.restype = C.c_size_t
    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ZSTD_CCtx_setPledgedSrcSize.argtypes = [C.c_void_p, C.c_ulonglong]
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, level)
    if checksum:
        L.ZSTD_CCtx_setParameter(cctx, 201, 1)
    if pledge:
        L.ZSTD_CCtx_setPledgedSrcSize(cctx, len(data))
    out = np.empty(1 << 20, dtype=np.uint8)
    pieces = []
    src = np.frombuffer(data, dtype=np.uint8) if data else np.empty(0, dtype=np.uint8)
    offs = list(range(0, len(data), piece)) or [0]
    for k, off in enumerate(offs):
        part = src[off:off + piece]
        inb = stream._InBuffer(part.ctypes.data if part.size else None, part.size, 0)
        mode = 2 if k == len(offs) - 1 else 0
        while True:
            outb = stream._OutBuffer(out.ctypes.data, out.size, 0)
            rem = L.ZSTD_compressStream2(cctx, C.byref(outb), C.byref(inb), mode)
            assert not L.ZSTD_isError(rem)
            pieces.append(out[:outb.pos].tobytes())
            if inb.pos == inb.size and (mode == 0 or rem == 0):
                break
    L.ZSTD_freeCCtx(cctx)
    return b"".join(pieces)


def ref_stream_decompress(ref, comp, piece=131702):
    """ZSTD_decompressStream of the reference library (src/raw-file-in.c:94-103)."""
    L = ref.L
    L.ZSTD_decompressStream.restype = C.c_size_t
    L.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    dctx = L.ZSTD_createDCtx()
    out = np.empty(1 << 20, dtype=np.uint8)
    src = np.frombuffer(comp, dtype=np.uint8)
    pieces = []
    for off in range(0, len(comp), piece):
        part = src[off:off + piece]
        inb = stream._InBuffer(part.ctypes.data, part.size, 0)
        while True:
            outb = stream._OutBuffer(out.ctypes.data, out.size, 0)
            r = L.ZSTD_decompressStream(dctx, C.byref(outb), C.byref(inb))
            assert not L.ZSTD_isError(r), L.ZSTD_getErrorName(r)
            pieces.append(out[:outb.pos].tobytes())
            if inb.pos == inb.size and outb.pos < outb.size:
                break
    L.ZSTD_freeDCtx(dctx)
    return b"".join(pieces)


@needs_ref
@pytest.mark.parametrize("n", [0, 1, 1000, 131072, 131073, 1 << 20, (9 << 20) + 12345, (17 << 20) + 1])
def test_stream_compress_is_read_by_the_reference(n, ref):
    data = GENERATORS["text"](n, seed=n & 0xFFF) if n else b""
    cctx = z.zstd_cctx()
    comp = stream.Compressor(cctx._ptr, pledged=n)
    parts = [comp.feed(data[off:off + 3000000]) for off in range(0, n, 3000000)]
    parts.append(comp.finish())
    blob = b"".join(parts)
    assert ref_stream_decompress(ref, blob) == data
    assert z.zstd_decompress(blob) == data
    # same bytes as the one-shot entry as long as the jobs line up with it (a job is a whole number of chunks)
    assert blob == z.zstd_compress(data)
    # the context serves the next session
    again = stream.Compressor(cctx._ptr).finish(data)
    assert again == blob


def test_stream_session_rules():
    L = _lib.lib()
    cctx = z.zstd_cctx(include_checksum=True)
    data = GENERATORS["iid10"](300000, seed=4)
    # a wrong pledge is an error at the end (zstd.c:24140)
    comp = stream.Compressor(cctx._ptr, pledged=len(data) + 1)
    comp.feed(data)
    with pytest.raises(stream.ZstdStreamError, match="Src size is incorrect"):
        comp.finish()
    # flush hands out what has been taken so far as complete frames
    comp = stream.Compressor(cctx._ptr)
    assert comp.feed(data[:100000]) == b""
    first = comp.flush()
    assert z.zstd_decompress(first) == data[:100000]
    rest = comp.finish(data[100000:])
    assert z.zstd_decompress(first + rest) == data
    # pledging in the middle of a session is refused (zstd.c:23795)
    comp = stream.Compressor(cctx._ptr)
    comp.feed(data[:10])
    assert _lib.is_error(L.zl_CCtx_setPledgedSrcSize(cctx._ptr, 5))
    L.zl_CCtx_reset(cctx._ptr, 1)
    # a tiny output buffer: the bytes come out over many calls, the return value counts down to 0
    src = np.frombuffer(data, dtype=np.uint8)
    inb = stream._InBuffer(src.ctypes.data, src.size, 0)
    small = np.empty(1000, dtype=np.uint8)
    got = []
    while True:
        outb = stream._OutBuffer(small.ctypes.data, small.size, 0)
        rem = L.zl_compressStream2(cctx._ptr, C.byref(outb), C.byref(inb), 2)
        assert not _lib.is_error(rem)
        got.append(small[:outb.pos].tobytes())
        if rem == 0:
            break
    assert inb.pos == inb.size and z.zstd_decompress(b"".join(got)) == data


@needs_ref
def test_stream_decompress_reads_the_reference(ref):
    dctx = z.zstd_dctx()
    for gen, n, kw in (("text", 3000000, {}), ("iid10", 1500000, {"pledge": True, "checksum": True}), ("int20", 400000, {"level": 9}),
                       ("text", 0, {}), ("rand", 200000, {"pledge": True})):
        data = GENERATORS[gen](n, seed=n & 0xFF) if n else b""
        comp = ref_stream_compress(ref, data, **kw)
        if n and not kw.get("pledge"):  # nothing pledged: the frame does not record its size (Single_Segment off, FCS flag 0)
            assert (comp[4] >> 6) == 0 and (comp[4] >> 5) & 1 == 0
        for piece in (131702, 1000, len(comp) or 1):
            dec = stream.Decompressor(dctx._ptr)
            out = b"".join(dec.feed(comp[off:off + piece]) for off in range(0, len(comp), piece))
            assert out == data, (gen, n, piece)
            assert dec.at_frame_boundary
    # a large frame that records no size still takes the large-frame pipeline (bounded by its block count)
    data = GENERATORS["text"](4000000, seed=77)
    comp = ref_stream_compress(ref, data)
    assert (comp[4] >> 6) == 0 and (comp[4] >> 5) & 1 == 0
    before = _lib.lib().zl_dctx_large_frames(dctx._ptr)
    dec = stream.Decompressor(dctx._ptr)
    assert b"".join(dec.feed(comp[off:off + 500000]) for off in range(0, len(comp), 500000)) == data
    assert _lib.lib().zl_dctx_large_frames(dctx._ptr) == before + 1
    out = (C.c_char * (len(data) + 100))()  # one-shot entry, more room than needed
    assert _lib.lib().zl_decompressDCtx(dctx._ptr, out, len(data) + 100, comp, len(comp)) == len(data) and out.raw[:len(data)] == data
    assert _lib.is_error(_lib.lib().zl_decompressDCtx(dctx._ptr, out, len(data) - 1, comp, len(comp)))  # too little room
    # concatenated streams, skippable frame in between, fed byte-odd pieces
    a, b = GENERATORS["text"](500000, seed=1), GENERATORS["iid10"](300000, seed=2)
    blob = ref_stream_compress(ref, a) + b"\x50\x2a\x4d\x18\x04\x00\x00\x00abcd" + z.zstd_compress(b) + ref.compress(a, level=1)
    dec = stream.Decompressor(dctx._ptr)
    out = b"".join(dec.feed(blob[off:off + 77777]) for off in range(0, len(blob), 77777))
    assert out == a + b + a and dec.at_frame_boundary
    # a stream that stops inside a frame is noticed; garbage is an error
    dec = stream.Decompressor(dctx._ptr)
    dec.feed(blob[:100000])
    assert not dec.at_frame_boundary
    with pytest.raises(stream.ZstdStreamError, match="Unknown frame descriptor"):
        stream.Decompressor(dctx._ptr).feed(b"this is not zstd at all, not even close")


def test_no_forward_progress_is_an_error():
    """zstd.c:42955-42962: sixteen calls that neither take input nor give output."""
    L = _lib.lib()
    dctx = z.zstd_dctx()
    comp = z.zstd_compress(GENERATORS["text"](100000, seed=1))
    src = np.frombuffer(comp, dtype=np.uint8)
    L.zl_DCtx_reset(dctx._ptr, 1)
    inb = stream._InBuffer(src.ctypes.data, src.size, 0)
    outb = stream._OutBuffer(None, 0, 0)  # no room at all
    codes = [L.zl_decompressStream(dctx._ptr, C.byref(outb), C.byref(inb)) for _ in range(20)]
    assert not _lib.is_error(codes[0]) and _lib.is_error(codes[-1])
    assert "output buffer being full" in _lib.error_name(codes[-1])


@needs_ref
def test_file_streaming_and_zstdfile(tmp_path, ref):
    """tests/testthat/test-compress-raw.R:38-105 and test-zstdfile.R: every writer's file is read by every reader."""
    data = GENERATORS["text"](2500000, seed=7)
    p1, p2, p3, p4 = (str(tmp_path / f"f{i}.zst") for i in range(4))
    z.zstd_compress(data, dst=p1, use_file_streaming=True)
    z.zstd_compress(data, dst=p2)
    with open(p3, "wb") as f:
        f.write(ref_stream_compress(ref, data, pledge=True))
    for p in (p1, p2, p3):
        assert z.zstd_decompress(p, use_file_streaming=True) == data
        assert z.zstd_decompress(p) == data
        with open(p, "rb") as f:
            assert ref_stream_decompress(ref, f.read()) == data
    obj = {"a": list(range(100000)), "b": "x" * 300000}
    z.zstd_serialize(obj, dst=p4, use_file_streaming=True, level=5)
    assert z.zstd_unserialize(p4, use_file_streaming=True) == obj
    assert z.zstd_unserialize(p4) == obj
    with pytest.raises(z.ZstdError, match="Couldn't open file"):
        z.zstd_decompress(str(tmp_path / "missing.zst"), use_file_streaming=True)
    # zstdfile: binary and text, written in pieces, read back whole, in pieces and by lines
    pb, pt = str(tmp_path / "b.zst"), str(tmp_path / "t.zst")
    with z.zstdfile(pb, "wb", level=1) as f:
        for off in range(0, len(data), 70001):
            f.write(data[off:off + 70001])
    with z.zstdfile(pb) as f:
        assert f.read(10) == data[:10]
        assert f.read() == data[10:]
        assert f.read() == b""
    with open(pb, "rb") as f:
        # the stream is ONE frame (a job's worth of input, 64 MiB, per frame): what a first-frame reader gets is everything
        assert ref.decompress(f.read()[: z.zstd_info(open(pb, "rb").read())["compressed_size"]], len(data)) == data
    lines = [f"line {i}: " + "é" * (i % 7) for i in range(50000)]
    with z.zstdfile(pt, "w") as f:
        f.writelines(lines)
    with z.zstdfile(pt, "r") as f:
        assert f.readlines() == lines
    with open(pt, "rb") as f:
        assert ref_stream_decompress(ref, f.read()).decode().splitlines() == lines
    with pytest.raises(ValueError):
        z.zstdfile(pt, "a")


def test_c_consumer_round_trips(tmp_path):
    """The C99 consumer of the header (tests/c/consumer.c), on the GPU: one-shot and streaming round trips from C."""
    import subprocess
    from test_abi import _build_c_consumer
    out = subprocess.run([_build_c_consumer(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "round trips ok" in out.stdout and out.stdout.strip().endswith("ok")


@needs_ref
def test_pledged_stream_above_one_job_is_one_frame(ref):
    """A stream whose size is pledged (what the reference's file writer does, src/raw-file-out.c:62) is ONE frame however
    many 64 MiB jobs it takes: the same bytes zl_compress2 writes for the whole input, so the reference's one-shot reader
    (first frame only, src/raw-file.c:150-155) gets all of it.  Without a pledge every job is a frame of its own."""
    import torch
    from zstdlite_amd.device import gen_mixed
    n = (150 << 20) + 4321
    data = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
    cctx = z.zstd_cctx()
    comp = stream.Compressor(cctx._ptr, pledged=n)
    blob = b"".join([comp.feed(data[off:off + 7000001]) for off in range(0, n, 7000001)] + [comp.finish()])
    assert ref.L.ZSTD_findFrameCompressedSize(blob, len(blob)) == len(blob) and ref.L.ZSTD_getFrameContentSize(blob, len(blob)) == n
    assert blob == z.zstd_compress(data)
    assert ref.decompress(blob, n) == data and z.zstd_decompress(blob) == data
    loose = stream.Compressor(cctx._ptr)
    blob2 = b"".join([loose.feed(data[off:off + 7000001]) for off in range(0, n, 7000001)] + [loose.finish()])
    assert ref.L.ZSTD_findFrameCompressedSize(blob2, len(blob2)) < len(blob2)  # three frames
    assert z.zstd_decompress(blob2) == data and ref_stream_decompress(ref, blob2) == data


@needs_ref
def test_flush_inside_a_pledged_one_frame_session_keeps_the_stream_valid(ref):
    """a pledged stream of more than one 64 MiB job is ONE frame whose runs end at job boundaries; a flush
    somewhere else must not put a frame of its own into the unterminated one.  It hands out the complete jobs and keeps
    collecting; the stream stays byte-identical to zl_compress2's and the reference reads it."""
    import torch
    from zstdlite_amd.device import gen_mixed
    n = (200 << 20) + 99
    data = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
    cctx = z.zstd_cctx()
    comp = stream.Compressor(cctx._ptr, pledged=n)
    cut = 74 << 20
    parts = [comp.feed(data[:cut]), comp.flush(), comp.feed(data[cut:cut + 1000]), comp.flush(), comp.flush(), comp.finish(data[cut + 1000:])]
    blob = b"".join(parts)
    # the first job runs beside the feed (round 4) and comes out with the feed or, at the latest, with the flush behind it;
    # the flushes after it cut nothing
    assert len(parts[0]) + len(parts[1]) > 0 and parts[3] == b"" and parts[4] == b""
    assert ref.L.ZSTD_findFrameCompressedSize(blob, len(blob)) == len(blob) and ref.L.ZSTD_getFrameContentSize(blob, len(blob)) == n
    assert blob == z.zstd_compress(data)
    assert ref.decompress(blob, n) == data
    # the flush that cannot release everything says so (zstd.h:841-880: non-zero = not fully flushed) -- once; repeated
    # without new input it returns 0, so that the documented "until it returns 0" loop ends.  An unpledged session's flush does
    # cut: everything written so far decodes from what it handed out.
    L = _lib.lib()
    hint = stream.Compressor(cctx._ptr, pledged=n)
    held = hint.feed(data[:cut])
    out = np.empty(1 << 20, dtype=np.uint8)
    inb = stream._InBuffer(None, 0, 0)
    rems = []
    for _ in range(400):  # (drain what is complete)
        outb = stream._OutBuffer(out.ctypes.data, out.size, 0)
        rem = L.zl_compressStream2(cctx._ptr, C.byref(outb), C.byref(inb), stream.E_FLUSH)
        assert not _lib.is_error(rem)
        held += out[:outb.pos].tobytes()
        rems.append((rem, outb.pos))
        if rem == 0:
            break
    assert rems[-1][0] == 0 and len(rems) < 400
    collected = cut - (64 << 20)  # one job is complete, the rest of the input stays collected
    assert any(r >= collected for r, _ in rems), rems[:4]
    assert len(held) > 0 and len(held) < len(blob)
    loose = stream.Compressor(cctx._ptr)
    first = loose.feed(data[:5 << 20]) + loose.flush()
    assert z.zstd_decompress(first) == data[:5 << 20] and ref_stream_decompress(ref, first) == data[:5 << 20]
    assert first + loose.finish(data[5 << 20:6 << 20]) != b""
    # more input than was pledged is an error, not a second frame (zstd.c:24140)
    over = stream.Compressor(cctx._ptr, pledged=1000)
    with pytest.raises(stream.ZstdStreamError, match="Src size is incorrect"):
        over.feed(data[:2000])


@needs_ref
@pytest.mark.parametrize("level", [9, -5, 19])
def test_streams_at_the_other_operating_points_write_what_the_one_shot_call_writes(ref, level):
    """A pledged stream's jobs are whole sections of ONE frame (above): at the deep levels (csrc/zl_deep.hip sorts a section at a time; the
    job's buffer holds its own range of the input only) and at the negative ones (stored literals) the bytes must still be zl_compress2's
    for the whole input, and the reference must read them; an unpledged session's frames too."""
    import torch
    from zstdlite_amd.device import gen_mixed
    n = (150 << 20) + 777
    data = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
    cctx = z.zstd_cctx(level=level)
    comp = stream.Compressor(cctx._ptr, pledged=n)
    blob = b"".join([comp.feed(data[off:off + 9999991]) for off in range(0, n, 9999991)] + [comp.finish()])
    assert blob == z.zstd_compress(data, level=level)
    assert ref.L.ZSTD_findFrameCompressedSize(blob, len(blob)) == len(blob)
    assert ref.decompress(blob, n) == data
    loose = stream.Compressor(cctx._ptr)
    blob2 = b"".join([loose.feed(data[off:off + 9999991]) for off in range(0, n, 9999991)] + [loose.finish()])
    assert ref_stream_decompress(ref, blob2) == data and z.zstd_decompress(blob2) == data


def test_a_job_whose_thread_cannot_be_started_loses_nothing():
    """when std::thread fails (EAGAIN) a full job runs inline; what it wrote used to REPLACE the bytes of the job
    before it, which collect() had just moved to where they wait for hand_out(): a run missing from a pledged one-frame stream, a whole
    frame from an unpledged one.  ZL_c_faultInjection = 2 << 4 lets the first job of a session have its thread and fails every later one."""
    import torch
    from zstdlite_amd.device import gen_mixed
    n = (200 << 20) + 4321
    data = gen_mixed(n, torch.device("cuda", 0))[0].cpu().numpy().tobytes()
    want = z.zstd_compress(data)
    cctx = z.zstd_cctx()
    assert _lib.lib().zl_CCtx_setParameter(cctx._ptr, 9098, 2 << 4) == 0
    # ONE call brings three jobs' worth of input: job 1 runs on its thread, job 2's thread "cannot be had" -- collect() has just put
    # job 1's bytes where they wait to be handed out, and job 2's must go behind them
    comp = stream.Compressor(cctx._ptr, pledged=n)
    blob = comp.feed(data) + comp.finish()
    assert blob == want, (len(blob), len(want))
    loose = stream.Compressor(cctx._ptr)
    blob2 = loose.feed(data) + loose.finish()
    assert z.zstd_decompress(blob2) == data


@needs_ref
def test_a_large_frame_is_decoded_while_it_arrives(ref, oracle):
    """Round 4 ("streaming decode of a large frame is not streaming"): zl_decompressStream used to collect a frame
    to its last byte before the first one came out.  A frame whose header records ZL_d_slabMiB (32 MiB in a stream) of content or
    more is now uploaded piece by piece, decoded in runs of blocks as they complete and handed out as the runs finish (zl_api.hip
    StreamFrame).  Here with ZL_d_slabMiB = 1 and frames of a few MiB: output must start before the frame's last piece has been
    fed, the bytes must be the input's, frames in front of and behind the large one (small ones, a skippable one) must come out
    in order, a damaged frame must end in the error the oracle names for the whole buffer, and the frame's content checksum is checked."""
    L = _lib.lib()
    rng = np.random.default_rng(17)
    text = GENERATORS["text"](4 << 20, seed=21)
    mixed = b"".join((text, GENERATORS["rand"](1 << 20, seed=22), GENERATORS["int20"](2 << 20, seed=23))[k % 3][(k * 33333) % (1 << 19):][: 160 << 10] for k in range(30))
    small = text[:70000]
    skippable = (0x184D2A53).to_bytes(4, "little") + (9).to_bytes(4, "little") + b"skip me!!"
    frames = {"ref3": ref.compress(mixed, level=3, checksum=True), "ref19": ref.compress(text, level=19), "own": z.zstd_compress(mixed, include_checksum=True),
              "own_text": z.zstd_compress(text)}
    data = {"ref3": mixed, "ref19": text, "own": mixed, "own_text": text}
    for name, f in frames.items():
        for piece in (1 << 20, 131702, 5000):
            dctx = z.zstd_dctx()
            assert L.zl_DCtx_setParameter(dctx._ptr, 9010, 1) == 0  # ZL_d_slabMiB
            dec = stream.Decompressor(dctx._ptr)
            before = int(L.zl_dctx_frame_runs(dctx._ptr))
            got, early, asked = [], 0, False
            for pos in range(0, len(f), piece):
                got.append(dec.feed(f[pos:pos + piece]))
                if not asked and pos + piece >= (len(f) * 3) // 5 and pos + piece < len(f):
                    # three fifths of the frame are in (at least one run of 256 KiB is complete): calls without input wait for what is
                    # in flight, so output must come now -- before the rest of the frame has been seen
                    asked = True
                    for _ in range(3):
                        got.append(dec.feed(b""))
                    early = sum(map(len, got))
            assert b"".join(got) == data[name] and dec.at_frame_boundary, (name, piece)
            assert int(L.zl_dctx_frame_runs(dctx._ptr)) - before >= 2, (name, piece)
            assert not asked or early > 0, (name, piece, "nothing came out before the frame was complete")
    # small frame, skippable frame, large frame, small frame: one stream, in order
    cat = ref.compress(small, level=3) + skippable + frames["ref3"] + z.zstd_compress(small) + frames["own_text"]
    want = small + mixed + small + text
    for piece in (300000, 4096):
        dctx = z.zstd_dctx()
        assert L.zl_DCtx_setParameter(dctx._ptr, 9010, 1) == 0
        dec = stream.Decompressor(dctx._ptr)
        assert b"".join(dec.feed(cat[pos:pos + piece]) for pos in range(0, len(cat), piece)) == want and dec.at_frame_boundary
    # a stream that stops inside the large frame is not at a frame boundary
    dctx = z.zstd_dctx()
    assert L.zl_DCtx_setParameter(dctx._ptr, 9010, 1) == 0
    dec = stream.Decompressor(dctx._ptr)
    part = dec.feed(frames["ref3"][: len(frames["ref3"]) // 2])
    assert not dec.at_frame_boundary and mixed.startswith(part)
    # damage: the verdict is the oracle's for the whole buffer (an error at whatever piece, or the same bytes)
    f = frames["ref3"]
    errs = 0
    for k in range(24):
        bad = bytearray(f)
        if k == 0:
            bad[-1] ^= 0x40  # the content checksum
        else:
            for _ in range(int(rng.integers(1, 3))):
                bad[int(rng.integers(8, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        try:
            want_bytes = oracle.decompress(bad, cap=len(mixed) + 16)
        except Exception:
            want_bytes = None
        dctx = z.zstd_dctx()
        assert L.zl_DCtx_setParameter(dctx._ptr, 9010, 1) == 0
        dec = stream.Decompressor(dctx._ptr)
        try:
            got_bytes = b"".join(dec.feed(bad[pos:pos + 200000]) for pos in range(0, len(bad), 200000))
            if not dec.at_frame_boundary:
                got_bytes = None
        except stream.ZstdStreamError:
            got_bytes = None
        assert (got_bytes is None) == (want_bytes is None), ("verdicts differ", k)
        assert got_bytes == want_bytes
        errs += got_bytes is None
    assert errs >= 8
