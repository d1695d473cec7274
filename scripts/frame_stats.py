"""Where the bytes of a zstd stream go: per frame set, the sum over all blocks of literal-section bytes,
sequence-section bytes, sequences and literals.  Used to compare the HIP encoder with the reference on
the same input (development / analysis; runs wherever the two libraries do).
usage: python scripts/frame_stats.py corpus [offset] [length]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def walk(stream):
    """-> dict of totals over every block of every frame in `stream`."""
    tot = dict(frames=0, blocks=0, raw_blocks=0, lit_sec=0, seq_sec=0, nseq=0, nlit=0, hdr=0, lit_raw=0, lit_huf=0, trees=0, tables=0)
    pos = 0
    n = len(stream)
    while pos < n:
        magic = int.from_bytes(stream[pos:pos + 4], "little")
        if (magic & 0xFFFFFFF0) == 0x184D2A50:
            pos += 8 + int.from_bytes(stream[pos + 4:pos + 8], "little")
            continue
        fhd = stream[pos + 4]
        single = (fhd >> 5) & 1; fcs = fhd >> 6; did = fhd & 3
        hs = 5 + (0 if single else 1) + (4 if did == 3 else did) + ((1 if single else 0) if fcs == 0 else (1 << fcs))
        tot["frames"] += 1
        tot["hdr"] += hs
        pos += hs
        while True:
            bh = int.from_bytes(stream[pos:pos + 3], "little")
            last, btype, bsize = bh & 1, (bh >> 1) & 3, bh >> 3
            pos += 3
            tot["hdr"] += 3
            tot["blocks"] += 1
            if btype != 2:
                tot["raw_blocks"] += 1
                csz = 1 if btype == 1 else bsize
                tot["nlit"] += bsize
                tot["lit_sec"] += csz
            else:
                blk = stream[pos:pos + bsize]
                t, sf = blk[0] & 3, (blk[0] >> 2) & 3
                if t < 2:
                    lh = [1, 2, 1, 3][sf]; regen = int.from_bytes(blk[:lh], "little") >> (3 if lh == 1 else 4)
                    lsec = lh + (regen if t == 0 else 1)
                    tot["lit_raw"] += regen
                else:
                    lh = [3, 3, 4, 5][sf]; bits = [10, 10, 14, 18][sf]
                    h = int.from_bytes(blk[:lh], "little"); regen = (h >> 4) & ((1 << bits) - 1); csz = h >> (4 + bits)
                    lsec = lh + csz
                    tot["lit_huf"] += regen
                    tot["trees"] += 1 if t == 2 else 0
                tot["nlit"] += regen
                tot["lit_sec"] += lsec
                tot["seq_sec"] += bsize - lsec
                s = blk[lsec:]
                ns = s[0]
                if ns >= 128:
                    ns = ((s[0] - 128) << 8) + s[1] if s[0] < 255 else s[1] + (s[2] << 8) + 0x7F00
                    mb = s[2] if s[0] < 255 else s[3]
                else:
                    mb = s[1] if ns else 0xFC
                tot["nseq"] += ns
                tot["tables"] += sum(1 for k in (6, 4, 2) if ((mb >> k) & 3) == 2)
                csz = bsize
            pos += csz
            if last:
                break
        if fhd & 4:
            pos += 4
    return tot


def show(name, stream, n):
    t = walk(stream)
    match_bytes = n - t["nlit"]
    print(f"{name:5s} total {len(stream):9d} ratio {n / len(stream):6.3f} | lit_sec {t['lit_sec']:8d} ({8 * t['lit_sec'] / max(t['nlit'], 1):5.2f} b/lit, {t['nlit']:8d} lits, raw {t['lit_raw']}) | "
          f"seq_sec {t['seq_sec']:8d} ({8 * t['seq_sec'] / max(t['nseq'], 1):5.2f} b/seq, {t['nseq']:7d} seqs, mean ML {match_bytes / max(t['nseq'], 1):5.2f}) | "
          f"hdr {t['hdr']} blocks {t['blocks']} raw {t['raw_blocks']} trees {t['trees']} fse tables {t['tables']}")
    return t


if __name__ == "__main__":
    import zstdlite_amd as z
    from zl_testlib import GENERATORS, LONG_RANGE, Ref
    name = sys.argv[1] if len(sys.argv) > 1 else "corpus"
    gen = LONG_RANGE.get(name) or GENERATORS[name]
    data = gen(None) if name == "corpus" else gen(8 << 20)
    off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ln = int(sys.argv[3]) if len(sys.argv) > 3 else len(data) - off
    data = data[off:off + ln]
    ref = Ref()
    show("ours", z.zstd_compress(data), len(data))
    show("ref", ref.compress(data, 3), len(data))
    chunks = b"".join(ref.compress(data[i:i + (128 << 10)], 3) for i in range(0, len(data), 128 << 10))
    show("ref/c", chunks, len(data))
