"""Shared test helpers: ctypes bindings to the oracle (oracle/libzl_oracle.so, this repo's
CPU restatement) and -- when present -- to the reference libzstd built by `make -C oracle ref`
(oracle/_ref/libzstd_ref.so), plus the synthetic input generators of SURVEY.md section 8(d).

TEST INFRASTRUCTURE ONLY: nothing under zstdlite_amd/ imports this module.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libzl_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libzstd_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_ERR_FLOOR = (1 << 64) - 120


def is_error(code):
    return code > _ERR_FLOOR


def _ensure_oracle():
    src = os.path.join(ORACLE_DIR, "zl_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libzl_oracle.so"], stdout=subprocess.DEVNULL)


class Oracle:
    """This repo's CPU restatement of the reference decode path (oracle/zl_oracle.c)."""

    def __init__(self):
        _ensure_oracle()
        L = C.CDLL(ORACLE_SO)
        L.zlo_decompress.restype = C.c_size_t
        L.zlo_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.zlo_find_frame_compressed_size.restype = C.c_size_t
        L.zlo_find_frame_compressed_size.argtypes = [C.c_void_p, C.c_size_t]
        L.zlo_get_frame_content_size.restype = C.c_ulonglong
        L.zlo_get_frame_content_size.argtypes = [C.c_void_p, C.c_size_t]
        L.zlo_xxh64.restype = C.c_uint64
        L.zlo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.zlo_error_name.restype = C.c_char_p
        L.zlo_error_name.argtypes = [C.c_size_t]
        L.zlo_compress_bound.restype = C.c_size_t
        L.zlo_compress_bound.argtypes = [C.c_size_t]
        L.zlo_dict_id.restype = C.c_uint
        L.zlo_dict_id.argtypes = [C.c_void_p, C.c_size_t]
        L.zlo_debug_first_block_sequences.restype = C.c_size_t
        L.zlo_debug_first_block_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        self.L = L

    def decompress(self, src, cap=None, dict=None, ignore_checksum=False):
        src = bytes(src)
        if cap is None:
            cap = 0
            pos = 0
            while pos < len(src):
                fs = self.L.zlo_find_frame_compressed_size(src[pos:], len(src) - pos)
                if is_error(fs):
                    break
                c = self.L.zlo_get_frame_content_size(src[pos:], len(src) - pos)
                if c >= (1 << 64) - 2:
                    raise ValueError("unknown content size: pass cap")
                cap += c
                pos += fs
        dst = C.create_string_buffer(max(cap, 1))
        d = bytes(dict) if dict is not None else None
        r = self.L.zlo_decompress(dst, cap, src, len(src), d, len(d) if d else 0, int(ignore_checksum))
        if is_error(r):
            raise RuntimeError(self.L.zlo_error_name(r).decode())
        return dst.raw[:r]

    def decompress_code(self, src, cap, dict=None, ignore_checksum=False):
        src = bytes(src)
        dst = C.create_string_buffer(max(cap, 1))
        d = bytes(dict) if dict is not None else None
        r = self.L.zlo_decompress(dst, cap, src, len(src), d, len(d) if d else 0, int(ignore_checksum))
        return r, dst.raw[: r if not is_error(r) else 0]

    def error_name(self, code):
        return self.L.zlo_error_name(code).decode()

    def xxh64(self, b, seed=0):
        b = bytes(b)
        return self.L.zlo_xxh64(b, len(b), seed)

    def first_block_sequences(self, src):
        src = bytes(src)
        cap = 50000
        arr = (C.c_uint32 * (3 * cap))()
        lit = C.c_size_t(0)
        n = self.L.zlo_debug_first_block_sequences(src, len(src), arr, cap, C.byref(lit))
        if is_error(n):
            raise RuntimeError(self.error_name(n))
        return np.frombuffer(arr, dtype=np.uint32)[: 3 * n].reshape(n, 3).copy(), lit.value


def have_ref():
    return os.path.exists(REF_SO)


class Ref:
    """The reference's bundled libzstd 1.5.6, compiled from /root/reference by oracle/Makefile."""

    ZSTD_c_compressionLevel = 100
    ZSTD_c_checksumFlag = 201
    ZSTD_c_nbWorkers = 400
    ZSTD_d_forceIgnoreChecksum = 1002  # ZSTD_d_experimentalParam3

    def __init__(self):
        L = C.CDLL(REF_SO)
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        L.ZSTD_CCtx_setParameter.restype = C.c_size_t
        L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_CCtx_loadDictionary.restype = C.c_size_t
        L.ZSTD_CCtx_loadDictionary.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ZSTD_compress2.restype = C.c_size_t
        L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_createDCtx.restype = C.c_void_p
        L.ZSTD_freeDCtx.argtypes = [C.c_void_p]
        L.ZSTD_DCtx_setParameter.restype = C.c_size_t
        L.ZSTD_DCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ZSTD_DCtx_loadDictionary.restype = C.c_size_t
        L.ZSTD_DCtx_loadDictionary.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ZSTD_decompressDCtx.restype = C.c_size_t
        L.ZSTD_decompressDCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_findFrameCompressedSize.restype = C.c_size_t
        L.ZSTD_findFrameCompressedSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
        L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_getErrorName.restype = C.c_char_p
        L.ZSTD_getErrorName.argtypes = [C.c_size_t]
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZDICT_trainFromBuffer.restype = C.c_size_t
        L.ZDICT_trainFromBuffer.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint]
        L.ZSTD_versionString.restype = C.c_char_p
        self.L = L

    def version(self):
        return self.L.ZSTD_versionString().decode()

    def compress(self, data, level=3, checksum=False, dict=None, nb_workers=0):
        """The reference's call sequence: init_cctx_with_opts (src/cctx.c:213) then
        ZSTD_compress2 into a ZSTD_compressBound buffer (src/raw-file.c:52-74)."""
        data = bytes(data)
        cctx = self.L.ZSTD_createCCtx()
        try:
            self.L.ZSTD_CCtx_setParameter(cctx, self.ZSTD_c_compressionLevel, level)
            if checksum:
                self.L.ZSTD_CCtx_setParameter(cctx, self.ZSTD_c_checksumFlag, 1)
            if nb_workers:
                self.L.ZSTD_CCtx_setParameter(cctx, self.ZSTD_c_nbWorkers, nb_workers)
            if dict is not None:
                d = bytes(dict)
                r = self.L.ZSTD_CCtx_loadDictionary(cctx, d, len(d))
                assert not self.L.ZSTD_isError(r), self.L.ZSTD_getErrorName(r)
            cap = self.L.ZSTD_compressBound(len(data))
            dst = C.create_string_buffer(cap)
            r = self.L.ZSTD_compress2(cctx, dst, cap, data, len(data))
            if self.L.ZSTD_isError(r):
                raise RuntimeError(self.L.ZSTD_getErrorName(r).decode())
            return dst.raw[:r]
        finally:
            self.L.ZSTD_freeCCtx(cctx)

    def decompress_code(self, src, cap, dict=None, ignore_checksum=False):
        src = bytes(src)
        dctx = self.L.ZSTD_createDCtx()
        try:
            if ignore_checksum:
                self.L.ZSTD_DCtx_setParameter(dctx, self.ZSTD_d_forceIgnoreChecksum, 1)
            if dict is not None:
                d = bytes(dict)
                self.L.ZSTD_DCtx_loadDictionary(dctx, d, len(d))
            dst = C.create_string_buffer(max(cap, 1))
            r = self.L.ZSTD_decompressDCtx(dctx, dst, cap, src, len(src))
            return r, dst.raw[: r if not self.L.ZSTD_isError(r) else 0]
        finally:
            self.L.ZSTD_freeDCtx(dctx)

    def decompress(self, src, cap, dict=None, ignore_checksum=False):
        r, out = self.decompress_code(src, cap, dict, ignore_checksum)
        if self.L.ZSTD_isError(r):
            raise RuntimeError(self.L.ZSTD_getErrorName(r).decode())
        return out

    def error_name(self, code):
        return self.L.ZSTD_getErrorName(code).decode()

    def train_dict(self, samples, dict_size):
        """src/dictionaries.c:78-233 -> ZDICT_trainFromBuffer (zstd.c:50979)."""
        flat = b"".join(samples)
        sizes = (C.c_size_t * len(samples))(*[len(s) for s in samples])
        buf = C.create_string_buffer(dict_size)
        r = self.L.ZDICT_trainFromBuffer(buf, dict_size, flat, sizes, len(samples))
        if r > _ERR_FLOOR:
            raise RuntimeError("ZDICT_trainFromBuffer failed")
        return buf.raw[:r]


# ----------------------------------------------------------------------------------------
# Synthetic inputs; fixed seeds, numpy only.
# ----------------------------------------------------------------------------------------
SEED = 0x9E3779B97F4A7C15

_VOCAB = [b"the", b"of", b"and", b"compression", b"zstd", b"frame", b"block", b"literal", b"sequence", b"offset",
          b"match", b"length", b"huffman", b"entropy", b"table", b"state", b"window", b"a", b"to", b"in", b"is",
          b"that", b"for", b"with", b"as", b"raw", b"vector", b"serialize", b"stream", b"dictionary", b"checksum",
          b"wavefront"]
assert len(_VOCAB) == 32


def gen_rand(n, seed=SEED):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def gen_iid10(n, seed=SEED):
    return np.random.default_rng(seed + 1).integers(1, 11, size=n, dtype=np.uint8).tobytes()


def gen_int20(n, seed=SEED):
    k = (n + 3) // 4
    return np.random.default_rng(seed + 2).integers(0, 1 << 20, size=k, dtype=np.uint32).astype("<u4").tobytes()[:n]


def gen_text(n, seed=SEED):
    rng = np.random.default_rng(seed + 3)
    lens = np.array([len(w) for w in _VOCAB], dtype=np.int64)
    flat = np.frombuffer(b"".join(_VOCAB), dtype=np.uint8)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    out = bytearray()
    while len(out) < n:
        k = 1 << 16
        idx = rng.integers(0, 32, size=k)
        sep = np.where(rng.integers(0, 8, size=k) == 0, 10, 32).astype(np.uint8)
        wl = lens[idx] + 1
        ends = np.cumsum(wl)
        total = int(ends[-1])
        word_of = np.repeat(np.arange(k), wl)
        pos_in = np.arange(total) - np.repeat(ends - wl, wl)
        is_sep = pos_in == np.repeat(wl - 1, wl)
        src_idx = np.repeat(starts[idx], wl) + np.minimum(pos_in, np.repeat(lens[idx] - 1, wl))
        chunk = flat[src_idx]
        chunk = np.where(is_sep, sep[word_of], chunk)
        out += chunk.tobytes()
    return bytes(out[:n])


GENERATORS = {"rand": gen_rand, "iid10": gen_iid10, "text": gen_text, "int20": gen_int20}


# ---- inputs with LONG-RANGE redundancy (repeats further back than one 128 KiB chunk): what the
# reference's level 3 sees through its 2 MiB window (zstd.c:29529) and chunk-local matching cannot
def gen_records(n, seed=SEED):
    """A 1 MiB pseudo-random block repeated, every copy after the first with 1 % of its bytes replaced."""
    rng = np.random.default_rng(seed + 4)
    blk = 1 << 20
    base = rng.integers(0, 256, size=blk, dtype=np.uint8)
    parts = []
    for i in range((n + blk - 1) // blk):
        b = base.copy()
        if i:
            idx = rng.integers(0, blk, size=blk // 100)
            b[idx] = rng.integers(0, 256, size=idx.size, dtype=np.uint8)
        parts.append(b)
    return np.concatenate(parts).tobytes()[:n] if parts else b""


_CORPUS_ROOT = "/usr/lib/python3.10"


def gen_corpus(n=None, seed=None):
    """The .py files of the Python standard library of this image (identical here and on the GPU box),
    sorted by path and concatenated (about 10.7 MB); n = None gives all of it, a larger n repeats it."""
    import glob
    key = "_corpus_cache"
    blob = globals().get(key)
    if blob is None:
        files = sorted(f for f in glob.glob(os.path.join(_CORPUS_ROOT, "**", "*.py"), recursive=True)
                       if "site-packages" not in f and "dist-packages" not in f)
        blob = b"".join(open(f, "rb").read() for f in files)
        globals()[key] = blob
    if n is None:
        return blob
    reps = -(-n // max(len(blob), 1))
    return (blob * reps)[:n]


LONG_RANGE = {"records": gen_records, "corpus": gen_corpus}


def sha256(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def load_golden():
    """tests/golden/frames.npz (made by tests/golden/make_golden.py from the reference libzstd).
    Returns (list of case dicts with 'comp' bytes and regenerated 'plain' bytes, dict bytes)."""
    import json
    z = np.load(os.path.join(GOLDEN, "frames.npz"))
    meta = json.loads(z["meta"].tobytes().decode())
    cases = []
    for m in meta:
        m = dict(m)
        m["comp"] = z[m["key"]].tobytes()
        if m["gen"] == "dict":
            m["plain"] = z[m["key"] + "_plain"].tobytes()
        else:
            m["plain"] = GENERATORS[m["gen"]](m["n"])
        assert sha256(m["plain"]) == m["sha256"], "generator drifted from the committed golden hashes"
        cases.append(m)
    return cases, z["dict"].tobytes()


def load_golden_large():
    """tests/golden/large_frames.npz (make_golden.py main_large): reference-made single frames of more
    than eight blocks.  Returns case dicts with 'comp' and the regenerated 'plain'."""
    import json
    z = np.load(os.path.join(GOLDEN, "large_frames.npz"))
    cases = []
    for m in json.loads(z["meta"].tobytes().decode()):
        m = dict(m)
        m["comp"] = z[m["key"]].tobytes()
        m["plain"] = b"".join(bytes(n) if g == "zeros" else GENERATORS[g](n, seed=seed) for g, n, seed in m["parts"])
        assert sha256(m["plain"]) == m["sha256"], "generator drifted from the committed golden hashes"
        cases.append(m)
    return cases
