"""BASELINE config C1 without R: the 1000 bytes R's serialize(head(mtcars), version = 3, xdr = TRUE) produces -- what
zstd_serialize() hands to ZSTD_compress2 (src/serialize.c:46-91) -- built by hand from the format (R Internals 1.8:
header, VECSXP with names / row.names / class attributes, eleven REALSXP columns).  The reference's README shows the
result of compressing exactly these bytes: 482 bytes, frame header 28 b5 2f fd 60 e8 02 (content size 0x02e8 + 256 =
1000), and its first 100 bytes (README.md:96-105) -- the known answer tests/test_oracle.py pins.  Writes
tests/golden/c1_mtcars_head.bin."""
import os
import struct

COLS = {
    "mpg": [21.0, 21.0, 22.8, 21.4, 18.7, 18.1], "cyl": [6, 6, 4, 6, 8, 6], "disp": [160, 160, 108, 258, 360, 225],
    "hp": [110, 110, 93, 110, 175, 105], "drat": [3.90, 3.90, 3.85, 3.08, 3.15, 2.76], "wt": [2.620, 2.875, 2.320, 3.215, 3.440, 3.460],
    "qsec": [16.46, 17.02, 18.61, 19.44, 17.02, 20.22], "vs": [0, 0, 1, 1, 0, 1], "am": [1, 1, 1, 0, 0, 0], "gear": [4, 4, 4, 3, 3, 3],
    "carb": [4, 4, 1, 1, 2, 1],
}
ROWS = ["Mazda RX4", "Mazda RX4 Wag", "Datsun 710", "Hornet 4 Drive", "Hornet Sportabout", "Valiant"]
README_FIRST_100 = bytes.fromhex(
    "28b52ffd60e802c50e0086134c4130676c3a00f17071267f4c13ec490775e964d95769c718a812f8c6205d6c2200d5a24bc8"
    "acdfc39967c3c55c31c3434a22401c688523acc7d1617e4d83c0827e0a380042003500242ec148db82a1b624622ce6c3b0a7")


def i32(v):
    return struct.pack(">i", v)


def charsxp(s):
    b = s.encode()
    return i32(0x00040009) + i32(len(b)) + b   # CHARSXP, ASCII


def strsxp(items):
    return i32(0x10) + i32(len(items)) + b"".join(charsxp(x) for x in items)


def attr(tag, value):
    return i32(0x402) + i32(0x01) + charsxp(tag) + value   # LISTSXP with tag; SYMSXP


def build(r_version=(4, 3, 1)):
    out = b"X\n" + i32(3) + i32((r_version[0] << 16) | (r_version[1] << 8) | r_version[2]) + i32(0x00030500) + i32(5) + b"UTF-8"
    out += i32(0x313) + i32(len(COLS))   # VECSXP, object bit, attributes
    for v in COLS.values():
        out += i32(0x0E) + i32(len(v)) + b"".join(struct.pack(">d", float(x)) for x in v)
    out += attr("names", strsxp(list(COLS))) + attr("row.names", strsxp(ROWS)) + attr("class", strsxp(["data.frame"])) + i32(0xFE)
    return out


if __name__ == "__main__":
    buf = build()
    assert len(buf) == 1000, len(buf)
    here = os.path.dirname(os.path.abspath(__file__))
    open(os.path.join(here, "c1_mtcars_head.bin"), "wb").write(buf)
    print("wrote", len(buf), "bytes")
