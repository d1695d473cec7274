// zl_blocks.h -- block-level header parsing shared by the lane pipeline (zl_decode2.hip) and the
// large-frame pipeline (zl_decode3.hip).
#pragma once
#include "zl_device.h"
#include "zl_format.h"
#include "zl_kernels.h"

// Literals and sequences section headers of one compressed block (zstd.c:43146-43230, :43707-43735)
// -> block record.  Returns 0 or an error code; lit_bytes: scratch the block's Huffman literals need.
__device__ __forceinline__ u32 parse_block_sections(const u8* blk, u32 bsize, zl2_blk_rec& r, u32& lit_bytes) {
    lit_bytes = 0;
    if (bsize > ZL_BLOCK_MAX) return ZL_E_SRCSIZE_WRONG;
    // literals section header (zstd.c:43146-43230)
    if (bsize < 1) return ZL_E_CORRUPTION;
    const u32 lt = blk[0] & 3, sf = (blk[0] >> 2) & 3;
    u32 lsec;
    if (lt < 2) {
        u32 lh, regen;
        if (sf == 0 || sf == 2) { lh = 1; regen = blk[0] >> 3; }
        else if (sf == 1) { if (bsize < 2) return ZL_E_CORRUPTION; lh = 2; regen = zl_rd_le(blk, 2) >> 4; }
        else { if (bsize < 3) return ZL_E_CORRUPTION; lh = 3; regen = zl_rd_le(blk, 3) >> 4; }
        if (regen > ZL_BLOCK_MAX) return ZL_E_CORRUPTION;
        r.lit_size = regen;
        if (lt == 0) {
            if (lh + regen > bsize) return ZL_E_CORRUPTION;
            r.lit_mode = 0; r.lit_a = lh; lsec = lh + regen;
        } else {
            if (lh + 1 > bsize) return ZL_E_CORRUPTION;
            r.lit_mode = 1; r.lit_a = blk[lh]; lsec = lh + 1;
        }
    } else {
        if (bsize < 5) return ZL_E_CORRUPTION;
        const u32 h4 = zl_rd_le(blk, 4);
        u32 lh, regen, cs, streams = 4;
        if (sf == 0 || sf == 1) { streams = sf == 0 ? 1 : 4; lh = 3; regen = (h4 >> 4) & 0x3FF; cs = (h4 >> 14) & 0x3FF; }
        else if (sf == 2) { lh = 4; regen = (h4 >> 4) & 0x3FFF; cs = h4 >> 18; }
        else { lh = 5; regen = (h4 >> 4) & 0x3FFFF; cs = (h4 >> 22) + ((u32)blk[4] << 10); }
        if (regen > ZL_BLOCK_MAX) return ZL_E_CORRUPTION;
        if (streams == 4 && regen < 6) return ZL_E_LITERALS_HEADER;
        if (lh + cs > bsize) return ZL_E_CORRUPTION;
        r.lit_mode = lt; r.lit_size = regen; r.lit_a = lh; r.lit_b = cs; r.streams = streams;
        lsec = lh + cs;
        lit_bytes = (regen + 3) & ~3u;
    }
    // sequences section header (zstd.c:43707-43735)
    const u8* sp = blk + lsec;
    const u32 sn = bsize - lsec;
    if (sn < 1) return ZL_E_SRCSIZE_WRONG;
    u32 nseq = sp[0], sh = 1;
    if (nseq > 0x7F) {
        if (nseq == 0xFF) { if (sn < 3) return ZL_E_SRCSIZE_WRONG; nseq = zl_rd_le(sp + 1, 2) + 0x7F00; sh = 3; }
        else { if (sn < 2) return ZL_E_SRCSIZE_WRONG; nseq = ((nseq - 0x80) << 8) + sp[1]; sh = 2; }
    }
    if (nseq == 0 && sh != sn) return ZL_E_CORRUPTION;
    r.nseq = nseq; r.seq_off = lsec + sh; r.seq_len = sn - sh;
    r.seq_modes = (nseq && sn > sh) ? sp[sh] : 0x100u;
    return 0;
}

