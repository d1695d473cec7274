// zl_decode.hip -- frame-parallel zstd decoder for gfx950 (MI355X).
//
// One 256-thread workgroup owns one frame at a time (persistent grid + atomic work queue)
// and walks its blocks in order, as ZSTD_decompressFrame does (zstd.c:41554-41660).  Inside
// a compressed block the three stages of ZSTD_decompressBlock_internal (zstd.c:45084) are
// re-shaped for a 64-wide-wavefront machine:
//
//   stage H  header parse + table build in LDS       (ZSTD_decodeLiteralsBlock zstd.c:43146,
//            HUF_readStats zstd.c:3470, HUF_readDTableX1 zstd.c:38436,
//            ZSTD_decodeSeqHeaders zstd.c:43707, FSE_readNCount zstd.c:3269,
//            ZSTD_buildFSETable zstd.c:43497)
//   stage A  Huffman literal decode, one lane per stream (wave 1)   } run concurrently
//   stage B  FSE sequence decode, one lane (wave 0), repcodes resolved (zstd.c:44241)  }
//   stage C  sequence execution by all 256 lanes: a block-wide prefix scan turns
//            (litLength, matchLength) into output offsets, then every 4-byte output word is
//            produced independently by chasing its source through the batch's sequence
//            table (literal buffer / earlier output) -- stores are coalesced dwords and no
//            lane waits on another lane's copy (replaces the serial ZSTD_execSequence,
//            zstd.c:44013).
//
// All arithmetic is integer; results must be bit-exact with the reference decoder.
#include <hip/hip_runtime.h>

#include "zl_device.h"
#include "zl_format.h"
#include "zl_kernels.h"
// Every workgroup barrier in this file is zl_sync_global(): this kernel hands literals, sequences and
// output between waves through global memory all over the place and is not the fast path.

#define DEC_THREADS 256
#define SEQ_BATCH 256

// ---------------------------------------------------------------------------------------
// constant tables (zstd.c:15462-15512): baselines / extra bits / default distributions
// ---------------------------------------------------------------------------------------
__constant__ u32 c_LL_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                  20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000,
                                  0x2000, 0x4000, 0x8000, 0x10000};
__constant__ u8 c_LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ u32 c_ML_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                  21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                  43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203, 0x403, 0x803, 0x1003, 0x2003,
                                  0x4003, 0x8003, 0x10003};
__constant__ u8 c_ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ short c_LL_default[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ short c_ML_default[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ short c_OF_default[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

// FSE sequence decode-table cell: [31:0] baseValue, [47:32] next-state base,
// [55:48] extra bits of the symbol, [63:56] state bits.  One 8-byte LDS read per state
// per sequence (same idea as ZSTD_seqSymbol, zstd.c:40088-40093).
__device__ __forceinline__ u64 seq_cell(u32 base_value, u32 next_base, u32 add_bits, u32 nb) {
    return (u64)base_value | ((u64)(next_base & 0xFFFF) << 32) | ((u64)add_bits << 48) | ((u64)nb << 56);
}

enum { K_LL = 0, K_OF = 1, K_ML = 2, K_WEIGHT = 3 };

struct DecShared {
    u64 fse_ll[512];
    u64 fse_ml[512];
    u64 fse_of[256];
    u16 huf[4096];  // [7:0] symbol, [15:8] code length; doubles as the XXH64 staging tile
    u8 weights[256];
    u8 cellsym[512];
    u16 fse_next[256];
    short norm[256];
    u32 wfse[64];  // Huffman-weight FSE table: [15:0] base, [23:16] nbits, [31:24] symbol
    u32 rank_start[16];
    // stage C batch tables
    u32 b_out[SEQ_BATCH + 1];
    u32 b_lit[SEQ_BATCH];
    u32 b_ll[SEQ_BATCH];
    u32 b_off[SEQ_BATCH];
    u32 scan_a[4], scan_b[4];
    u64 xxh_acc[4];
    // frame / block control (written by thread 0, read by all after a barrier)
    int err;
    u32 frame;
    u32 ll_log, ml_log, of_log, huf_log;
    int tables_valid, huf_valid;
    u32 rep[3];
    u32 hdr_size;
    u64 fcs;
    int has_checksum, skippable;
    u32 blk_type, blk_size, blk_last;
    u32 lit_mode;  // 0 = bytes at lit_ptr, 1 = RLE byte
    u32 lit_rle;
    u32 lit_size;
    u32 lit_streams;
    u32 lit_from_src;  // literals are raw bytes inside the compressed block
    u32 lit_src_off;   // offset in block of raw literals / Huffman streams
    u32 lit_csize;     // size of Huffman streams (after tree description)
    u32 nseq;
    u32 seq_off, seq_len;  // FSE bitstream inside the block
    u64 stamp_a;           // profiling: cycles the literal lanes spent (max over lanes)
    u32 dict_live;         // which of this workgroup's tables still hold the dictionary's: bit 0 fse_ll, 1 fse_ml, 2 fse_of, 3 huf
};

// ---------------------------------------------------------------------------------------
// FSE table description (zstd.c:3269 FSE_readNCount_body).  Forward LSB-first bit reads.
// Runs on one thread.  Returns bytes consumed, or 0 on error (a description is never empty).
// ---------------------------------------------------------------------------------------
__device__ u32 read_ncount(short* norm, int* max_sym_io, int* log_out, int max_log, const u8* src, u32 n) {
    if (n < 1) return 0;
    i32 bit = 0;
    const i32 total = (i32)n * 8;
    int log = (int)zl_bits_fwd(src, n, bit, 4) + 5;
    bit += 4;
    if (log > max_log) return 0;
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1;
    int sym = 0;
    const int limit = *max_sym_io;
    bool prev_zero = false;
    for (int s = 0; s <= limit; s++) norm[s] = 0;
    while (remaining > 1 && sym <= limit) {
        if (prev_zero) {
            for (;;) {
                if (bit + 2 > total) return 0;
                int r = (int)zl_bits_fwd(src, n, bit, 2);
                bit += 2;
                sym += r;
                if (r != 3) break;
            }
            if (sym > limit) return 0;
            prev_zero = false;
        }
        if (bit > total) return 0;
        int max = (2 * threshold - 1) - remaining;
        u32 peek = zl_bits_fwd(src, n, bit, nbits);
        int count;
        if ((int)(peek & (u32)(threshold - 1)) < max) {
            count = (int)(peek & (u32)(threshold - 1));
            bit += nbits - 1;
        } else {
            count = (int)(peek & (u32)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            bit += nbits;
        }
        count--;
        remaining -= (count < 0) ? -count : count;
        norm[sym++] = (short)count;
        prev_zero = (count == 0);
        while (remaining < threshold && threshold > 1) {
            nbits--;
            threshold >>= 1;
        }
    }
    if (remaining != 1 || bit > total) return 0;
    *max_sym_io = sym - 1;
    *log_out = log;
    return (u32)((bit + 7) >> 3);
}

// zstd.c:3690-3790 / zstd.c:43497-43600: spread + per-cell state transition. One thread.
// kind selects the cell format.  Returns false on a malformed distribution.
__device__ bool build_fse(DecShared& S, const short* norm, int max_sym, int log, int kind) {
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s <= max_sym; s++) {
        if (norm[s] == -1) {
            S.cellsym[high--] = (u8)s;
            S.fse_next[s] = 1;
        } else {
            S.fse_next[s] = (u16)norm[s];
        }
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s <= max_sym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            S.cellsym[pos] = (u8)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (int u = 0; u < size; u++) {
        u32 s = S.cellsym[u];
        u32 ns = S.fse_next[s]++;
        u32 nb = (u32)log - (31 - __clz(ns));
        u32 base = (ns << nb) - (u32)size;
        switch (kind) {
        case K_LL: S.fse_ll[u] = seq_cell(c_LL_base[s], base, c_LL_bits[s], nb); break;
        case K_ML: S.fse_ml[u] = seq_cell(c_ML_base[s], base, c_ML_bits[s], nb); break;
        case K_OF: S.fse_of[u] = seq_cell(1u << s, base, s, nb); break;
        default: S.wfse[u] = (base & 0xFFFF) | (nb << 16) | (s << 24); break;
        }
    }
    return true;
}

// One of predefined / RLE / compressed / repeat (zstd.c:43640 ZSTD_buildSeqTable).
// Returns bytes consumed or 0xFFFFFFFF on error.
__device__ u32 read_seq_table(DecShared& S, int kind, int mode, const u8* src, u32 n) {
    const int max_sym = kind == K_LL ? ZL_MAX_LL : (kind == K_ML ? ZL_MAX_ML : ZL_MAX_OF);
    const int max_log = kind == K_OF ? ZL_OF_LOG_MAX : ZL_LL_LOG_MAX;
    u32* log_slot = kind == K_LL ? &S.ll_log : (kind == K_ML ? &S.ml_log : &S.of_log);
    switch (mode) {
    case 0: {
        const short* def = kind == K_LL ? c_LL_default : (kind == K_ML ? c_ML_default : c_OF_default);
        int dmax = kind == K_LL ? 35 : (kind == K_ML ? 52 : 28), dlog = kind == K_OF ? 5 : 6;
        for (int s = 0; s <= dmax; s++) S.norm[s] = def[s];
        if (!build_fse(S, S.norm, dmax, dlog, kind)) return 0xFFFFFFFFu;
        *log_slot = (u32)dlog;
        return 0;
    }
    case 1: {
        if (n < 1) return 0xFFFFFFFFu;
        u32 s = src[0];
        if ((int)s > max_sym) return 0xFFFFFFFFu;
        if (kind == K_LL) S.fse_ll[0] = seq_cell(c_LL_base[s], 0, c_LL_bits[s], 0);
        else if (kind == K_ML) S.fse_ml[0] = seq_cell(c_ML_base[s], 0, c_ML_bits[s], 0);
        else S.fse_of[0] = seq_cell(1u << s, 0, s, 0);
        *log_slot = 0;
        return 1;
    }
    case 2: {
        int ms = max_sym, log = 0;
        u32 used = read_ncount(S.norm, &ms, &log, max_log, src, n);
        if (used == 0) return 0xFFFFFFFFu;
        if (!build_fse(S, S.norm, ms, log, kind)) return 0xFFFFFFFFu;
        *log_slot = (u32)log;
        return used;
    }
    default: return S.tables_valid ? 0 : 0xFFFFFFFFu;
    }
}

// Huffman tree description -> weights[] (zstd.c:3470 HUF_readStats_body; FSE-compressed
// weights via zstd.c:3808 FSE_decompress_usingDTable_generic).  One thread.
// Returns bytes consumed, 0 on error.  Sets S.huf_log and S.rank_start[].
__device__ u32 read_huf_weights(DecShared& S, const u8* src, u32 n) {
    if (n < 1) return 0;
    int nsym = 0;
    u32 used;
    const u32 hb = src[0];
    if (hb >= 128) {
        nsym = (int)hb - 127;
        used = 1 + (u32)(nsym + 1) / 2;
        if (used > n) return 0;
        for (int i = 0; i < nsym; i++) {
            u8 b = src[1 + i / 2];
            S.weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
    } else {
        used = 1 + hb;
        if (used > n || hb < 2) return 0;
        int max_sym = 255, log = 0;
        u32 hs = read_ncount(S.norm, &max_sym, &log, 6, src + 1, hb);
        if (hs == 0 || hs >= hb) return 0;
        if (!build_fse(S, S.norm, max_sym, log, K_WEIGHT)) return 0;
        const u8* bs = src + 1 + hs;
        const u32 bn = hb - hs;
        i32 pos = zl_bwd_init(bs, bn);
        if (pos < 0) return 0;
        pos -= log;
        u32 s1 = zl_bits_bwd(bs, bn, pos, log);
        pos -= log;
        u32 s2 = zl_bits_bwd(bs, bn, pos, log);
        if (pos < 0) return 0;
        for (;;) {
            if (nsym > 253) return 0;
            u32 e = S.wfse[s1];
            S.weights[nsym++] = (u8)(e >> 24);
            u32 nb = (e >> 16) & 0xFF;
            pos -= (i32)nb;
            s1 = (e & 0xFFFF) + zl_bits_bwd(bs, bn, pos, nb);
            if (pos < 0) {
                S.weights[nsym++] = (u8)(S.wfse[s2] >> 24);
                break;
            }
            if (nsym > 253) return 0;
            e = S.wfse[s2];
            S.weights[nsym++] = (u8)(e >> 24);
            nb = (e >> 16) & 0xFF;
            pos -= (i32)nb;
            s2 = (e & 0xFFFF) + zl_bits_bwd(bs, bn, pos, nb);
            if (pos < 0) {
                S.weights[nsym++] = (u8)(S.wfse[s1] >> 24);
                break;
            }
        }
    }
    u32 total = 0;
    u32 rank[16];
    for (int w = 0; w < 16; w++) rank[w] = 0;
    for (int i = 0; i < nsym; i++) {
        u32 w = S.weights[i];
        if (w > ZL_HUF_LOG_MAX) return 0;
        rank[w]++;
        if (w) total += 1u << (w - 1);
    }
    if (total == 0) return 0;
    const u32 log = 32 - __clz(total);  // highbit(total) + 1
    if (log > ZL_HUF_LOG_MAX) return 0;
    const u32 rest = (1u << log) - total;
    if (rest & (rest - 1)) return 0;
    const u32 last_w = 32 - __clz(rest);
    S.weights[nsym++] = (u8)last_w;
    rank[last_w]++;
    if (rank[1] < 2 || (rank[1] & 1)) return 0;  // zstd.c:3549
    for (int i = nsym; i < 256; i++) S.weights[i] = 0;
    u32 cell = 0;
    for (u32 w = 1; w <= log; w++) {
        S.rank_start[w] = cell;
        cell += rank[w] << (w - 1);
    }
    S.huf_log = log;
    return used;
}

// All threads: fill the single-symbol decode table (zstd.c:38436 HUF_readDTableX1_wksp):
// ascending weight, ascending symbol; a symbol of weight w owns 2^(w-1) consecutive cells.
__device__ void fill_huf_table(DecShared& S) {
    const u32 t = threadIdx.x;  // symbol value
    const u32 w = S.weights[t];
    if (w) {
        u32 before = 0;
        for (u32 s = 0; s < t; s++) before += (S.weights[s] == w);
        const u32 cnt = 1u << (w - 1);
        const u32 start = S.rank_start[w] + before * cnt;
        const u16 e = (u16)(t | ((S.huf_log + 1 - w) << 8));
        for (u32 k = 0; k < cnt; k++) S.huf[start + k] = e;
    }
}

// Stage A: one backward Huffman stream, one lane (zstd.c:38560 HUF_decompress1X1 body).
__device__ bool huf_decode_stream(const DecShared& S, u8* dst, u32 nout, const u8* src, u32 n) {
    i32 pos = zl_bwd_init(src, n);
    if (pos < 0) return false;
    const u32 log = S.huf_log;
    zl_bwd_reader r;
    zl_bwd_open(r, src, n, pos);
    u32 i = 0;
    // four symbols per iteration, packed into one dword store when the destination allows
    const u32 head = (u32)((4 - ((uintptr_t)dst & 3)) & 3);
    for (; i < head && i < nout; i++) {
        const u32 e = S.huf[zl_bwd_read(r, pos - (i32)log, log)];
        dst[i] = (u8)e;
        pos -= (i32)(e >> 8);
    }
    for (; i + 4 <= nout; i += 4) {
        u32 word = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 e = S.huf[zl_bwd_read(r, pos - (i32)log, log)];
            word |= (e & 0xFF) << (8 * j);
            pos -= (i32)(e >> 8);
        }
        *(u32*)(dst + i) = word;
    }
    for (; i < nout; i++) {
        const u32 e = S.huf[zl_bwd_read(r, pos - (i32)log, log)];
        dst[i] = (u8)e;
        pos -= (i32)(e >> 8);
    }
    return pos == 0;  // zstd.c:38640 BIT_endOfDStream
}

// Stage B: sequence decode, one lane (zstd.c:44241-44356 ZSTD_decodeSequence, repcode
// rules zstd.c:44291-44327).  Writes resolved (ll, ml, off) triples.
__device__ int decode_sequences(DecShared& S, const u8* bs, u32 bn, u32 nseq, u32* o_ll, u32* o_ml, u32* o_off) {
    i32 bp = zl_bwd_init(bs, bn);
    if (bp < 0) return ZL_E_CORRUPTION;
    zl_bwd_reader r;
    zl_bwd_open(r, bs, bn, bp);
    bp -= (i32)S.ll_log;
    u32 sl = zl_bwd_read(r, bp, S.ll_log);
    bp -= (i32)S.of_log;
    u32 so = zl_bwd_read(r, bp, S.of_log);
    bp -= (i32)S.ml_log;
    u32 sm = zl_bwd_read(r, bp, S.ml_log);
    if (bp < 0) return ZL_E_CORRUPTION;
    u32 r0 = S.rep[0], r1 = S.rep[1], r2 = S.rep[2];
    for (u32 i = 0; i < nseq; i++) {
        const u64 el = S.fse_ll[sl], eo = S.fse_of[so], em = S.fse_ml[sm];
        u32 nb = (u32)(eo >> 48) & 0xFF;
        bp -= (i32)nb;
        const u32 ofv = (u32)eo + zl_bwd_read(r, bp, nb);
        nb = (u32)(em >> 48) & 0xFF;
        bp -= (i32)nb;
        const u32 ml = (u32)em + zl_bwd_read(r, bp, nb);
        nb = (u32)(el >> 48) & 0xFF;
        bp -= (i32)nb;
        const u32 ll = (u32)el + zl_bwd_read(r, bp, nb);
        if (bp < 0) return ZL_E_CORRUPTION;
        u32 off;
        if (ofv > 3) {
            off = ofv - 3;
            r2 = r1; r1 = r0; r0 = off;
        } else {
            const u32 idx = ofv - 1 + (ll == 0);
            if (idx == 0) {
                off = r0;
            } else {
                off = (idx == 3) ? r0 - 1 : (idx == 1 ? r1 : r2);
                if (off == 0) return ZL_E_CORRUPTION;
                if (idx != 1) r2 = r1;
                r1 = r0;
                r0 = off;
            }
        }
        o_ll[i] = ll;
        o_ml[i] = ml;
        o_off[i] = off;
        if (i + 1 < nseq) {
            nb = (u32)(el >> 56);
            bp -= (i32)nb;
            sl = ((u32)(el >> 32) & 0xFFFF) + zl_bwd_read(r, bp, nb);
            nb = (u32)(em >> 56);
            bp -= (i32)nb;
            sm = ((u32)(em >> 32) & 0xFFFF) + zl_bwd_read(r, bp, nb);
            nb = (u32)(eo >> 56);
            bp -= (i32)nb;
            so = ((u32)(eo >> 32) & 0xFFFF) + zl_bwd_read(r, bp, nb);
            if (bp < 0) return ZL_E_CORRUPTION;
        }
    }
    if (bp != 0) return ZL_E_CORRUPTION;  // zstd.c:44690
    S.rep[0] = r0; S.rep[1] = r1; S.rep[2] = r2;
    return 0;
}

// Block-wide exclusive scan of two u32 values (256 threads = 4 waves).
__device__ __forceinline__ void block_scan2(DecShared& S, u32 a, u32 b, u32& ex_a, u32& ex_b, u32& tot_a, u32& tot_b) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 ta = __shfl_up(ia, d, 64), tb = __shfl_up(ib, d, 64);
        if ((int)lane >= d) { ia += ta; ib += tb; }
    }
    if (lane == 63) { S.scan_a[wave] = ia; S.scan_b[wave] = ib; }
    zl_sync_global();
    u32 pa = 0, pb = 0, ta = 0, tb = 0;
#pragma unroll
    for (u32 w = 0; w < 4; w++) {
        u32 xa = S.scan_a[w], xb = S.scan_b[w];
        if (w < wave) { pa += xa; pb += xb; }
        ta += xa; tb += xb;
    }
    ex_a = pa + ia - a;
    ex_b = pb + ib - b;
    tot_a = ta;
    tot_b = tb;
    zl_sync_global();
}

struct LitSrc {
    const u8* ptr;
    u32 rle;  // 0x100 | byte when RLE
};
__device__ __forceinline__ u8 lit_byte(const LitSrc& L, u32 idx) { return L.rle ? (u8)L.rle : L.ptr[idx]; }

// Source of output byte p (block-relative) for the current batch: follow match references
// until they land in the literal buffer or in bytes produced before this batch.
__device__ __forceinline__ u8 resolve_byte(const DecShared& S, u32 nb, u32 p, u32 k, u32 batch_start,
                                            const u8* blk_out /* block start in dst */, i64 blk_abs /* frame pos of block */,
                                            const LitSrc& L, const u8* dict_end) {
    for (;;) {
        const u32 rel = p - S.b_out[k];
        const u32 ll = S.b_ll[k];
        if (rel < ll) return lit_byte(L, S.b_lit[k] + rel);
        const u32 m = rel - ll, o = S.b_off[k];
        i64 q = (m < o) ? (i64)p - (i64)o : (i64)(S.b_out[k] + ll) - (i64)o + (i64)(m % o);
        if (q < (i64)batch_start) {
            if (blk_abs + q >= 0) return blk_out[q];
            return dict_end[blk_abs + q];
        }
        p = (u32)q;
        // locate the sequence that produced p
        u32 lo = 0, hi = k + 1;
        while (hi - lo > 1) {
            u32 mid = (lo + hi) >> 1;
            if (S.b_out[mid] <= p) lo = mid; else hi = mid;
        }
        k = lo;
    }
}

// ---------------------------------------------------------------------------------------
// thread-0 header parsers
// ---------------------------------------------------------------------------------------
// zstd.c:41050-41148 ZSTD_getFrameHeader_advanced
__device__ int parse_frame_header(DecShared& S, const u8* src, u64 n, u32 dict_id) {
    S.skippable = 0;
    if (n < 5) return ZL_E_SRCSIZE_WRONG;
    const u32 magic = zl_rd_le(src, 4);
    if ((magic & 0xFFFFFFF0u) == ZL_MAGIC_SKIPPABLE) {
        S.skippable = 1;
        return 0;
    }
    if (magic != ZL_MAGIC) return ZL_E_PREFIX_UNKNOWN;
    const u32 fhd = src[4];
    const u32 dflag = fhd & 3, single = (fhd >> 5) & 1, fflag = fhd >> 6;
    if (fhd & 8) return ZL_E_FRAMEPARAM_UNSUPPORTED;
    const u32 dbytes = dflag == 3 ? 4 : dflag;
    const u32 fbytes = fflag == 0 ? (single ? 1 : 0) : (1u << fflag);
    const u32 hs = 5 + (single ? 0 : 1) + dbytes + fbytes;
    if (n < hs) return ZL_E_SRCSIZE_WRONG;
    u32 pos = 5;
    if (!single) {
        const u32 wlog = (src[pos++] >> 3) + 10;
        if (wlog > 31) return ZL_E_WINDOW_TOO_LARGE;
    }
    const u32 did = zl_rd_le(src + pos, dbytes);
    pos += dbytes;
    u64 fcs = ZL_CONTENTSIZE_UNKNOWN;
    if (fflag == 0) { if (single) fcs = src[pos]; }
    else if (fflag == 1) fcs = (u64)zl_rd_le(src + pos, 2) + 256;
    else if (fflag == 2) fcs = zl_rd_le(src + pos, 4);
    else fcs = (u64)zl_rd_le(src + pos, 4) | ((u64)zl_rd_le(src + pos + 4, 4) << 32);
    if (did && did != dict_id) return ZL_E_DICT_WRONG;  // zstd.c:41318
    S.fcs = fcs;
    S.has_checksum = (fhd >> 2) & 1;
    S.hdr_size = hs;
    return 0;
}

// zstd.c:43146-43351 ZSTD_decodeLiteralsBlock (header part).  blk = compressed block bytes.
// Returns bytes of the literals section, or 0 with S.err set.
__device__ u32 parse_literals(DecShared& S, const u8* blk, u32 n) {
    if (n < 1) { S.err = ZL_E_CORRUPTION; return 0; }
    const u32 type = blk[0] & 3, sf = (blk[0] >> 2) & 3;
    S.lit_from_src = 0;
    if (type < 2) {
        u32 hs, regen;
        if (sf == 0 || sf == 2) { hs = 1; regen = blk[0] >> 3; }
        else if (sf == 1) { if (n < 2) { S.err = ZL_E_CORRUPTION; return 0; } hs = 2; regen = zl_rd_le(blk, 2) >> 4; }
        else { if (n < 3) { S.err = ZL_E_CORRUPTION; return 0; } hs = 3; regen = zl_rd_le(blk, 3) >> 4; }
        if (regen > ZL_BLOCK_MAX) { S.err = ZL_E_CORRUPTION; return 0; }
        S.lit_size = regen;
        if (type == 0) {
            if (hs + regen > n) { S.err = ZL_E_CORRUPTION; return 0; }
            S.lit_mode = 0; S.lit_from_src = 1; S.lit_src_off = hs;
            return hs + regen;
        }
        if (hs + 1 > n) { S.err = ZL_E_CORRUPTION; return 0; }
        S.lit_mode = 1; S.lit_rle = blk[hs];
        return hs + 1;
    }
    if (n < 5) { S.err = ZL_E_CORRUPTION; return 0; }
    const u32 h4 = zl_rd_le(blk, 4);
    u32 hs, regen, csize, streams = 4;
    if (sf == 0 || sf == 1) { streams = sf == 0 ? 1 : 4; hs = 3; regen = (h4 >> 4) & 0x3FF; csize = (h4 >> 14) & 0x3FF; }
    else if (sf == 2) { hs = 4; regen = (h4 >> 4) & 0x3FFF; csize = h4 >> 18; }
    else { hs = 5; regen = (h4 >> 4) & 0x3FFFF; csize = (h4 >> 22) + ((u32)blk[4] << 10); }
    if (regen > ZL_BLOCK_MAX) { S.err = ZL_E_CORRUPTION; return 0; }
    if (streams == 4 && regen < 6) { S.err = ZL_E_LITERALS_HEADER; return 0; }
    if (hs + csize > n) { S.err = ZL_E_CORRUPTION; return 0; }
    u32 tree = 0;
    if (type == 3) {
        if (!S.huf_valid) { S.err = ZL_E_DICT_CORRUPTED; return 0; }
    } else {
        tree = read_huf_weights(S, blk + hs, csize);
        if (tree == 0) { S.err = ZL_E_CORRUPTION; return 0; }
    }
    S.lit_mode = (type == 3) ? 3 : 2;  // 2: table to be filled, 3: reuse
    S.lit_size = regen;
    S.lit_streams = streams;
    S.lit_src_off = hs + tree;
    S.lit_csize = csize - tree;
    if (streams == 4 && S.lit_csize < 10) { S.err = ZL_E_CORRUPTION; return 0; }
    return hs + csize;
}

// zstd.c:43707-43786 ZSTD_decodeSeqHeaders
__device__ void parse_seq_header(DecShared& S, const u8* p, u32 n) {
    if (n < 1) { S.err = ZL_E_SRCSIZE_WRONG; return; }
    u32 pos = 0;
    u32 nseq = p[pos++];
    if (nseq > 0x7F) {
        if (nseq == 0xFF) {
            if (n < 3) { S.err = ZL_E_SRCSIZE_WRONG; return; }
            nseq = zl_rd_le(p + 1, 2) + 0x7F00; pos = 3;
        } else {
            if (n < 2) { S.err = ZL_E_SRCSIZE_WRONG; return; }
            nseq = ((nseq - 0x80) << 8) + p[1]; pos = 2;
        }
    }
    S.nseq = nseq;
    if (nseq == 0) {
        if (pos != n) S.err = ZL_E_CORRUPTION;
        return;
    }
    if (nseq > ZL_SEQ_CAP) { S.err = ZL_E_CORRUPTION; return; }
    if (pos + 1 > n) { S.err = ZL_E_SRCSIZE_WRONG; return; }
    const u32 modes = p[pos++];
    if (modes & 3) { S.err = ZL_E_CORRUPTION; return; }
    if ((modes >> 6) != 3) S.dict_live &= ~1u;  // (anything but Repeat_Mode writes the table)
    if (((modes >> 2) & 3) != 3) S.dict_live &= ~2u;
    if (((modes >> 4) & 3) != 3) S.dict_live &= ~4u;
    u32 r = read_seq_table(S, K_LL, modes >> 6, p + pos, n - pos);
    if (r == 0xFFFFFFFFu) { S.err = ZL_E_CORRUPTION; return; }
    pos += r;
    r = read_seq_table(S, K_OF, (modes >> 4) & 3, p + pos, n - pos);
    if (r == 0xFFFFFFFFu) { S.err = ZL_E_CORRUPTION; return; }
    pos += r;
    r = read_seq_table(S, K_ML, (modes >> 2) & 3, p + pos, n - pos);
    if (r == 0xFFFFFFFFu) { S.err = ZL_E_CORRUPTION; return; }
    pos += r;
    S.tables_valid = 1;
    S.seq_off = pos;
    S.seq_len = n - pos;
}

// ---------------------------------------------------------------------------------------
// Stage C for one compressed block.  All threads.  Returns regenerated size via S (uniform).
// ---------------------------------------------------------------------------------------
__device__ u32 execute_block(DecShared& S, u8* blk_out, u64 cap_left, i64 blk_abs, const LitSrc& L,
                             const u32* q_ll, const u32* q_ml, const u32* q_off, const u8* dict_end, u32 dict_len) {
    const u32 t = threadIdx.x;
    const u32 nseq = S.nseq, lit_size = S.lit_size;
    u32 lit_pos = 0, out_pos = 0;  // uniform running totals (block-relative)
    for (u32 b0 = 0; b0 < nseq; b0 += SEQ_BATCH) {
        const u32 nb = (nseq - b0 < SEQ_BATCH) ? (nseq - b0) : SEQ_BATCH;
        u32 ll = 0, ml = 0, off = 1;
        if (t < nb) { ll = q_ll[b0 + t]; ml = q_ml[b0 + t]; off = q_off[b0 + t]; }
        u32 ex_l, ex_o, tot_l, tot_o;
        block_scan2(S, ll, ll + ml, ex_l, ex_o, tot_l, tot_o);
        if (t < nb) {
            const u32 my_out = out_pos + ex_o, my_lit = lit_pos + ex_l;
            S.b_out[t] = my_out;
            S.b_lit[t] = my_lit;
            S.b_ll[t] = ll;
            S.b_off[t] = off;
            // zstd.c:44030-44045 bounds: literals inside the literal buffer, match source inside
            // window+dictionary
            if ((u64)my_lit + ll > lit_size) S.err = ZL_E_CORRUPTION;
            if ((u64)off > (u64)(blk_abs + my_out + ll) + dict_len) S.err = ZL_E_CORRUPTION;
        }
        if (t == 0) S.b_out[nb] = out_pos + tot_o;
        const u32 batch_start = out_pos, batch_end = out_pos + tot_o;
        if ((u64)batch_end > cap_left || batch_end > ZL_BLOCK_MAX) {
            if (t == 0) S.err = (batch_end > ZL_BLOCK_MAX) ? ZL_E_CORRUPTION : ZL_E_DSTSIZE_TOO_SMALL;
        }
        zl_sync_global();
        if (S.err) return 0;
        // 4-byte output words aligned on the destination address
        const u32 mis = (u32)((uintptr_t)(blk_out + batch_start) & 3);
        const i64 w0 = (i64)batch_start - mis;
        for (i64 ws = w0 + 4 * (i64)t; ws < (i64)batch_end; ws += 4 * DEC_THREADS) {
            const u32 first = (ws < (i64)batch_start) ? batch_start : (u32)ws;
            const u32 last = ((u32)(ws + 4) > batch_end) ? batch_end : (u32)(ws + 4);
            // sequence owning `first`
            u32 lo = 0, hi = nb;
            while (hi - lo > 1) {
                u32 mid = (lo + hi) >> 1;
                if (S.b_out[mid] <= first) lo = mid; else hi = mid;
            }
            u32 k = lo, word = 0;
            for (u32 p = first; p < last; p++) {
                while (p >= S.b_out[k + 1]) k++;
                u8 v = resolve_byte(S, nb, p, k, batch_start, blk_out, blk_abs, L, dict_end);
                word |= (u32)v << (8 * (p - (u32)ws));
            }
            if (last - first == 4) {
                *(u32*)(blk_out + ws) = word;
            } else {
                for (u32 p = first; p < last; p++) blk_out[p] = (u8)(word >> (8 * (p - (u32)ws)));
            }
        }
        lit_pos += tot_l;
        out_pos += tot_o;
        zl_sync_global();  // batch output visible to the next batch's match reads
    }
    // trailing literals (zstd.c:44693-44700)
    const u32 rest = lit_size - lit_pos;
    if ((u64)out_pos + rest > cap_left) { if (t == 0) S.err = ZL_E_DSTSIZE_TOO_SMALL; }
    else if (out_pos + rest > ZL_BLOCK_MAX) { if (t == 0) S.err = ZL_E_CORRUPTION; }
    zl_sync_global();
    if (S.err) return 0;
    for (u32 i = t; i < rest; i += DEC_THREADS) blk_out[out_pos + i] = lit_byte(L, lit_pos + i);
    zl_sync_global();
    return out_pos + rest;
}

__global__ __launch_bounds__(DEC_THREADS, ZL_DEC_WGS_PER_CU) void zl_decode_kernel(zl_decode_args a) {
    __shared__ DecShared S;
    const u32 t = threadIdx.x;
    const u32 slot = blockIdx.x;
    u8* lit_buf = a.lit_scratch + (size_t)slot * ZL_LIT_STRIDE;
    u32* q_ll = a.seq_scratch + (size_t)slot * 3 * ZL_SEQ_CAP;
    u32* q_ml = q_ll + ZL_SEQ_CAP;
    u32* q_off = q_ml + ZL_SEQ_CAP;
    const zl_dict_tables* DT = (const zl_dict_tables*)a.dict_tables;
    const u8* dict_end = a.dict_content ? a.dict_content + a.dict_len : nullptr;

    // behind the lane pipeline: nothing to do unless it has counted frames for this kernel (word 9 of the counter block)
    if (a.plan && a.counter[9] == 0) return;
    if (t == 0) S.dict_live = 0;
    for (;;) {
        zl_sync_global();
        if (t == 0) S.frame = atomicAdd(a.counter, 1u);
        zl_sync_global();
        const u32 f = S.frame;
        if (f >= a.nframes) break;
        if (a.plan && a.plan[f].route != 0 && a.frec[f].err != ZL_E_FALLBACK) continue;  // decoded by zl_decode2.hip
        const zl_frame_desc fd = a.frames[f];
        const u8* src = a.src + fd.src_off;
        u8* out = a.dst + fd.dst_off;
        const u64 n = fd.src_size, cap = fd.dst_cap;

        if (t == 0) {
            S.err = parse_frame_header(S, src, n, a.dict_id);
            S.rep[0] = 1; S.rep[1] = 4; S.rep[2] = 8;
            S.tables_valid = 0; S.huf_valid = 0;
            if (DT && !S.err) {
                S.rep[0] = DT->rep[0]; S.rep[1] = DT->rep[1]; S.rep[2] = DT->rep[2];
                S.ll_log = DT->ll_log; S.ml_log = DT->ml_log; S.of_log = DT->of_log; S.huf_log = DT->huf_log;
                S.tables_valid = 1; S.huf_valid = 1;
            }
        }
        zl_sync_global();
        if (S.err || S.skippable) {
            if (t == 0) a.results[f] = S.err ? ZL_ERR(S.err) : 0;
            continue;
        }
        if (DT) {
            // (only the tables that a frame since the last copy has replaced: a batch of small frames that all ride on the
            // dictionary's tables -- what a dictionary is for -- copies them once per workgroup, not 18 KiB per frame)
            const u32 live = S.dict_live;
            if (!(live & 1)) for (u32 i = t; i < 512; i += DEC_THREADS) S.fse_ll[i] = DT->fse_ll[i];
            if (!(live & 2)) for (u32 i = t; i < 512; i += DEC_THREADS) S.fse_ml[i] = DT->fse_ml[i];
            if (!(live & 4)) for (u32 i = t; i < 256; i += DEC_THREADS) S.fse_of[i] = DT->fse_of[i];
            if (!(live & 8)) for (u32 i = t; i < 4096; i += DEC_THREADS) S.huf[i] = DT->huf[i];
            zl_sync_global();
            if (t == 0) S.dict_live = 15u;
        }
        u64 pos = S.hdr_size, op = 0;
        bool done = false;
        while (!done) {
            // ---- block header (zstd.c:43075 ZSTD_getcBlockSize)
            if (t == 0) {
                if (n - pos < 3) S.err = ZL_E_SRCSIZE_WRONG;
                else {
                    const u32 bh = zl_rd_le(src + pos, 3);
                    S.blk_last = bh & 1; S.blk_type = (bh >> 1) & 3; S.blk_size = bh >> 3;
                    const u32 csize = (S.blk_type == 1) ? 1 : S.blk_size;
                    if (S.blk_type == 3) S.err = ZL_E_CORRUPTION;
                    else if (n - pos - 3 < csize) S.err = ZL_E_SRCSIZE_WRONG;
                    else if (S.blk_type != 2 && S.blk_size > cap - op) S.err = ZL_E_DSTSIZE_TOO_SMALL;
                    else if (S.blk_type == 2 && S.blk_size > ZL_BLOCK_MAX) S.err = ZL_E_SRCSIZE_WRONG;
                }
            }
            zl_sync_global();
            if (S.err) break;
            pos += 3;
            const u32 btype = S.blk_type, bsize = S.blk_size;
            const u8* blk = src + pos;
            if (btype == 0) {
                for (u32 i = t; i < bsize; i += DEC_THREADS) out[op + i] = blk[i];
                op += bsize; pos += bsize;
            } else if (btype == 1) {
                const u8 v = blk[0];
                for (u32 i = t; i < bsize; i += DEC_THREADS) out[op + i] = v;
                op += bsize; pos += 1;
            } else {
                // ---- stage H: headers + tables
                u64 ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
                if (a.stats && t == 0) { ts0 = __builtin_amdgcn_s_memtime(); S.stamp_a = 0; }
                if (t == 0) {
                    const u32 lsec = parse_literals(S, blk, bsize);
                    if (!S.err) parse_seq_header(S, blk + lsec, bsize - lsec);
                    if (!S.err) S.seq_off += lsec;
                }
                zl_sync_global();
                if (S.err) break;
                if (S.lit_mode == 2) {
                    fill_huf_table(S);
                    if (t == 0) { S.huf_valid = 1; S.dict_live &= ~8u; }
                    zl_sync_global();
                }
                // ---- stages A + B, concurrently on different waves
                LitSrc L;
                L.rle = (S.lit_mode == 1) ? (0x100u | S.lit_rle) : 0;
                L.ptr = S.lit_from_src ? (blk + S.lit_src_off) : lit_buf;
                if (a.stats && t == 0) ts1 = __builtin_amdgcn_s_memtime();
                if (t == 0 && S.nseq) {
                    int e = decode_sequences(S, blk + S.seq_off, S.seq_len, S.nseq, q_ll, q_ml, q_off);
                    if (e) S.err = e;
                }
                if (a.stats && t == 0) ts2 = __builtin_amdgcn_s_memtime();
                const u64 ta0 = (a.stats && S.lit_mode >= 2 && t >= 64 && t < 68) ? __builtin_amdgcn_s_memtime() : 0;
                if (S.lit_mode >= 2) {
                    const u8* hs = blk + S.lit_src_off;
                    const u32 regen = S.lit_size, cs = S.lit_csize;
                    if (S.lit_streams == 1) {
                        if (t == 64) {
                            if (!huf_decode_stream(S, lit_buf, regen, hs, cs)) S.err = ZL_E_CORRUPTION;
                        }
                    } else if (t >= 64 && t < 68) {
                        const u32 j = t - 64;
                        const u32 s1 = zl_rd_le(hs, 2), s2 = zl_rd_le(hs + 2, 2), s3 = zl_rd_le(hs + 4, 2);
                        const u32 seg = (regen + 3) / 4;
                        if (6 + s1 + s2 + s3 > cs || seg * 3 > regen) {
                            S.err = ZL_E_CORRUPTION;
                        } else {
                            const u32 s4 = cs - 6 - s1 - s2 - s3;
                            const u32 so = j == 0 ? 0 : (j == 1 ? s1 : (j == 2 ? s1 + s2 : s1 + s2 + s3));
                            const u32 sl = j == 0 ? s1 : (j == 1 ? s2 : (j == 2 ? s3 : s4));
                            const u32 cnt = (j == 3) ? regen - 3 * seg : seg;
                            if (!huf_decode_stream(S, lit_buf + j * seg, cnt, hs + 6 + so, sl)) S.err = ZL_E_CORRUPTION;
                        }
                    }
                }
                if (a.stats && S.lit_mode >= 2 && t >= 64 && t < 68) atomicMax((unsigned long long*)&S.stamp_a, (unsigned long long)(__builtin_amdgcn_s_memtime() - ta0));
                zl_sync_global();
                if (S.err) break;
                if (a.stats && t == 0) ts3 = __builtin_amdgcn_s_memtime();
                // ---- stage C
                const u32 produced = execute_block(S, out + op, cap - op, (i64)op, L, q_ll, q_ml, q_off, dict_end, a.dict_len);
                if (a.stats && t == 0) {
                    const u64 ts4 = __builtin_amdgcn_s_memtime();
                    atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)(ts1 - ts0));  // H: headers + tables
                    atomicAdd((unsigned long long*)&a.stats[1], (unsigned long long)(ts2 - ts1));  // B: FSE chain
                    atomicAdd((unsigned long long*)&a.stats[2], (unsigned long long)S.stamp_a);    // A: slowest Huffman lane
                    atomicAdd((unsigned long long*)&a.stats[3], (unsigned long long)(ts3 - ts1));  // A||B wall
                    atomicAdd((unsigned long long*)&a.stats[4], (unsigned long long)(ts4 - ts3));  // C: execution
                    atomicAdd((unsigned long long*)&a.stats[5], 1ull);                            // compressed blocks
                    atomicAdd((unsigned long long*)&a.stats[6], (unsigned long long)S.nseq);
                    atomicAdd((unsigned long long*)&a.stats[7], (unsigned long long)(S.lit_mode >= 2 ? S.lit_size : 0));
                }
                if (S.err) break;
                op += produced; pos += bsize;
            }
            done = S.blk_last != 0;
            zl_sync_global();
        }
        zl_sync_global();
        if (!S.err) {
            if (t == 0) {
                if (S.fcs != ZL_CONTENTSIZE_UNKNOWN && S.fcs != op) S.err = ZL_E_CORRUPTION;  // zstd.c:41646
                else if (S.has_checksum && n - pos < 4) S.err = ZL_E_CHECKSUM_WRONG;
            }
            zl_sync_global();
            if (!S.err && S.has_checksum && !a.ignore_checksum) {
                const u32 got = zl_xxh64_low32((u8*)S.huf, S.xxh_acc, S.scan_a, out, op);
                if (t == 0) S.dict_live &= ~8u;  // (the Huffman table was the staging tile)
                if (t == 0 && got != zl_rd_le(src + pos, 4)) S.err = ZL_E_CHECKSUM_WRONG;  // zstd.c:41655
            }
            zl_sync_global();
        }
        if (t == 0) a.results[f] = S.err ? ZL_ERR(S.err) : op;
    }
}

// ---------------------------------------------------------------------------------------
// Dictionary digest (zstd.c:42053-42150 ZSTD_loadDEntropy): one workgroup parses the
// formatted dictionary once per context and leaves ready-to-copy tables in HBM.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void zl_digest_dict_kernel(const u8* dict, u32 n, zl_dict_tables* out) {
    __shared__ DecShared S;
    const u32 t = threadIdx.x;
    if (t == 0) {
        S.err = 0;
        S.tables_valid = 0;
        u32 pos = 8;
        u32 used = read_huf_weights(S, dict + pos, n - pos);
        if (used == 0) S.err = ZL_E_DICT_CORRUPTED;
        pos += used;
        S.hdr_size = pos;
    }
    zl_sync_global();
    if (!S.err) fill_huf_table(S);
    zl_sync_global();
    if (t == 0 && !S.err) {
        u32 pos = S.hdr_size;
        const int kinds[3] = {K_OF, K_ML, K_LL};
        out->sec[0] = 8;
        for (int i = 0; i < 3 && !S.err; i++) {
            out->sec[1 + i] = pos;
            u32 r = read_seq_table(S, kinds[i], 2, dict + pos, n - pos);
            if (r == 0xFFFFFFFFu) S.err = ZL_E_DICT_CORRUPTED;
            else pos += r;
        }
        out->sec[4] = pos;
        if (!S.err && pos + 12 > n) S.err = ZL_E_DICT_CORRUPTED;
        if (!S.err) {
            const u32 content = n - (pos + 12);
            for (int i = 0; i < 3; i++) {
                u32 rep = zl_rd_le(dict + pos + 4 * i, 4);
                if (rep == 0 || rep > content) S.err = ZL_E_DICT_CORRUPTED;
                out->rep[i] = rep;
            }
            out->content_off = pos + 12;
            out->ll_log = S.ll_log; out->ml_log = S.ml_log; out->of_log = S.of_log; out->huf_log = S.huf_log;
        }
        out->status = (u32)S.err;
    }
    zl_sync_global();
    if (!S.err) {
        for (u32 i = t; i < 512; i += DEC_THREADS) { out->fse_ll[i] = S.fse_ll[i]; out->fse_ml[i] = S.fse_ml[i]; }
        for (u32 i = t; i < 256; i += DEC_THREADS) out->fse_of[i] = S.fse_of[i];
        for (u32 i = t; i < 4096; i += DEC_THREADS) out->huf[i] = S.huf[i];
    }
}

// ---------------------------------------------------------------------------------------
// host-callable launchers (used by zl_api.hip)
// ---------------------------------------------------------------------------------------
extern "C" hipError_t zl_launch_decode(const zl_decode_args* a, u32 nslots, hipStream_t stream) {
    if (a->nframes == 0) return hipSuccess;
    u32 grid = a->nframes < nslots ? a->nframes : nslots;
    hipLaunchKernelGGL(zl_decode_kernel, dim3(grid), dim3(DEC_THREADS), 0, stream, *a);
    return hipGetLastError();
}

extern "C" hipError_t zl_launch_digest_dict(const u8* d_dict, u32 n, void* d_tables, hipStream_t stream) {
    hipLaunchKernelGGL(zl_digest_dict_kernel, dim3(1), dim3(DEC_THREADS), 0, stream, d_dict, n, (zl_dict_tables*)d_tables);
    return hipGetLastError();
}
