"""Inputs of the small-size ratio sweep (scripts/ratio_sweep.py, tests/test_compress_gpu.py::test_ratio_sweep_small_sizes):
what zstd_serialize() of ordinary R objects hands to the compressor, from 1 KB to 1 MB.  TEST INFRASTRUCTURE."""
import os
import pickle

import numpy as np

from zl_testlib import GENERATORS, GOLDEN

SIZES = [1 << 10, 4 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20]


def _float64_normal(n):
    """rnorm(): doubles with full-entropy mantissas (what man/benchmark.R's numeric columns hold)."""
    return np.random.default_rng(41).normal(size=(n + 7) // 8).astype("<f8").tobytes()[:n]


def _float64_rounded(n):
    """round(rnorm(), 2) * 100 stored as doubles: measurements with few significant digits -- most mantissa bytes are zero."""
    return np.round(np.random.default_rng(42).normal(size=(n + 7) // 8) * 100).astype("<f8").tobytes()[:n]


def _float64_seq(n):
    """seq(0, by = 0.25): a regular grid, the x column of a time series."""
    return (np.arange((n + 7) // 8, dtype="<f8") * 0.25).tobytes()[:n]


def _c1(n):
    """Config C1's buffer -- the R-native serialization of head(mtcars), 1000 bytes (tests/golden/make_c1.py) -- repeated
    with a counter between the copies: a list of similar small data frames."""
    buf = open(os.path.join(GOLDEN, "c1_mtcars_head.bin"), "rb").read()
    out = bytearray()
    i = 0
    while len(out) < n:
        out += buf + i.to_bytes(4, "little")
        i += 1
    return bytes(out[:n])


def _pickled_frame(n):
    """A data frame as a dict of columns (ints, rounded doubles, short strings, a factor-like column), pickled."""
    rng = np.random.default_rng(43)
    rows = max(4, n // 40)
    frame = {
        "id": list(range(rows)),
        "value": [round(float(x), 2) for x in rng.normal(50, 10, rows)],
        "count": [int(x) for x in rng.poisson(7, rows)],
        "label": [f"item_{int(x):04d}" for x in rng.integers(0, 500, rows)],
        "group": [("control", "treated", "placebo")[int(x)] for x in rng.integers(0, 3, rows)],
    }
    b = pickle.dumps(frame, protocol=4)
    while len(b) < n:
        b += b
    return b[:n]


INPUTS = {
    "text": lambda n: GENERATORS["text"](n, seed=n + 1),
    "iid10": lambda n: GENERATORS["iid10"](n, seed=n + 1),
    "int20": lambda n: GENERATORS["int20"](n, seed=n + 1),
    "f64_normal": _float64_normal,
    "f64_rounded": _float64_rounded,
    "f64_seq": _float64_seq,
    "c1_frames": _c1,
    "pickled_frame": _pickled_frame,
}
