// zl_decode2.hip -- lane-parallel decode pipeline for frames of up to eight blocks (<= 1 MiB of
// content): the shape of BASELINE config C2 (10k independent 64 KiB frames) and of everything the
// HIP encoder emits.  The same kernels decode the SEGMENTS of one large frame for the large-frame
// pipeline (zl_decode3.hip).  zl_decode.hip's one-workgroup-per-frame kernel stays as the general
// path (dictionaries, unusual tables); frames are routed per frame.
//
// The reference decodes a block with three serial loops (zstd.c:45084 ZSTD_decompressBlock_internal:
// Huffman literals zstd.c:38653, FSE sequences zstd.c:44241, sequence execution zstd.c:44013).
// Measured on MI355X, running those chains on one lane of a wavefront costs ~500 cycles per
// literal and ~2200 cycles per sequence while the other 63 lanes idle.  Here every lane of a
// wavefront runs its own chain, so a wave instruction advances many Huffman streams (or FSE
// chains) at once:
//
//   zl2_index_kernel  lane per frame: frame header, block walk, literal/sequence section headers
//                     (zstd.c:41050, :43075, :43146, :43707) -> block records
//   zl2_order_kernel  the frames sorted by decreasing sequence count / literal bytes: the schedule
//   zl2_lit_kernel    lane per (frame, block, Huffman stream): 8 frames x 8 blocks x 4 streams per
//                     workgroup; a block that brings a tree leads the treeless blocks behind it
//                     (zstd.c:38436); LDS rings keep memory off the decode chain
//   zl2_seq_kernel    lane per (frame, block): FSE table build + the 3-state sequence chain with
//                     branch-free repcode rules (zstd.c:43497, :44241); a block decoded before its
//                     predecessor carries its starting repcodes as symbols
//   zl2_entropy1_kernel  both roles in one launch when every frame has one block
//   zl2_exec_kernel   512-thread workgroup per frame; only the 8 KiB batch being built lives in LDS:
//                     literals scattered, every match byte one hop through the batch's sequence
//                     table, 16-bit links shortened by pointer jumping, 16-byte coalesced flush
//                     (replaces zstd.c:44013 ZSTD_execSequence); XXH64 check.  <true>: one run of
//                     segments of a large frame, bytes whose source lies in front of the run deferred
//
// All integer work; output must be bit-exact with the reference decoder.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdlib>

#include "zl_device.h"
#include "zl_format.h"
#include "zl_kernels.h"
#include "zl_blocks.h"

// -DZL_ROLE_STAMPS=1: cycle stamps inside the literal and sequence roles (the "entropy roles" line of
// scripts/kernel_bench.py).  Compiled out by default: even unused they cost the roles 1-2 %.
#ifndef ZL_ROLE_STAMPS
#define ZL_ROLE_STAMPS 0
#endif

// ---------------------------------------------------------------------------------------
// tables (zstd.c:15462-15512)
// ---------------------------------------------------------------------------------------
__constant__ u32 k_LL_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                  20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000,
                                  0x2000, 0x4000, 0x8000, 0x10000};
__constant__ u8 k_LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ u32 k_ML_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                  21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                  43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203, 0x403, 0x803, 0x1003, 0x2003,
                                  0x4003, 0x8003, 0x10003};
__constant__ u8 k_ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ short k_LL_default[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ short k_ML_default[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ short k_OF_default[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

// ---------------------------------------------------------------------------------------
// zl2_index_kernel: one lane per frame
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void set_err(zl2_frame_rec* fr, u32 code) {
    if (fr->err == 0) fr->err = code;
}

// Inherited table descriptions (zl_kernels.h zl2_args::ext): four records per segment of a large frame, or ONE set of four for
// every frame of a batch that starts from a dictionary (ext_shared), whose offsets count from ext_base.
__device__ __forceinline__ const zl2_blk_rec& ext_rec(const zl2_args& a, u32 f, u32 k) { return a.ext[(a.ext_shared ? (size_t)0 : (size_t)f * 4) + k]; }
// (a record whose pad[0] is set describes the context's DICTIONARY -- the synthetic block of zl_api.hip dctx_prepare_dict -- among a
// large frame's own donors: its offsets count from dict_ext_base)
__device__ __forceinline__ const u8* ext_src(const zl2_args& a, u32 f, const zl2_blk_rec& r) { return r.pad[0] ? a.dict_ext_base : (a.ext_base ? a.ext_base : a.src + a.frames[f].src_off); }
// does a chain of frame / segment f start from a history that the exec kernel fills in (symbols instead of numbers)?
__device__ __forceinline__ bool starts_symbolic(const zl2_args& a, u32 f) { return a.ext != nullptr && (f != 0 || a.ext_shared != 0 || a.sym_first != 0); }

__global__ __launch_bounds__(64) void zl2_index_kernel(zl2_args a) {
    const u32 f = blockIdx.x * 64 + threadIdx.x;
    if (f >= a.nframes) return;
    zl2_frame_rec* fr = a.frec + f;
    fr->err = 0;
    fr->nblocks = 0;
    fr->skippable = 0;
    fr->cost = 0;
    fr->lit_cost = 0;
    if (a.plan[f].route != 1) {
        fr->err = a.plan[f].route == 3 ? ZL_E_DONE : ZL_E_FALLBACK;
        if (a.plan[f].route == 0) atomicAdd(a.max_blocks + 1, 1u);  // (the word behind max_blocks: frames the general kernel has to take)
        return;
    }
    const zl_frame_desc fd = a.frames[f];
    const u8* src = a.src + fd.src_off;
    const u64 n = fd.src_size;
    // ---- frame header (zstd.c:41050-41148)
    if (n < 5) { fr->err = ZL_E_SRCSIZE_WRONG; return; }
    const u32 magic = zl_rd_le(src, 4);
    if ((magic & 0xFFFFFFF0u) == ZL_MAGIC_SKIPPABLE) { fr->skippable = 1; return; }
    if (magic != ZL_MAGIC) { fr->err = ZL_E_PREFIX_UNKNOWN; return; }
    const u32 fhd = src[4];
    const u32 dflag = fhd & 3, single = (fhd >> 5) & 1, fflag = fhd >> 6;
    if (fhd & 8) { fr->err = ZL_E_FRAMEPARAM_UNSUPPORTED; return; }
    const u32 dbytes = dflag == 3 ? 4 : dflag;
    const u32 fbytes = fflag == 0 ? (single ? 1 : 0) : (1u << fflag);
    const u32 hs = 5 + (single ? 0 : 1) + dbytes + fbytes;
    if (n < hs) { fr->err = ZL_E_SRCSIZE_WRONG; return; }
    u32 pos = 5;
    if (!single) {
        const u32 wlog = (src[pos++] >> 3) + 10;
        if (wlog > 31) { fr->err = ZL_E_WINDOW_TOO_LARGE; return; }
    }
    const u32 did = zl_rd_le(src + pos, dbytes);
    pos += dbytes;
    u64 fcs = ZL_CONTENTSIZE_UNKNOWN;
    if (fflag == 0) { if (single) fcs = src[pos]; }
    else if (fflag == 1) fcs = (u64)zl_rd_le(src + pos, 2) + 256;
    else if (fflag == 2) fcs = zl_rd_le(src + pos, 4);
    else fcs = (u64)zl_rd_le(src + pos, 4) | ((u64)zl_rd_le(src + pos + 4, 4) << 32);
    if (did && did != a.dict_id) { fr->err = ZL_E_DICT_WRONG; return; }  // zstd.c:41318 (a frame that names no dictionary still starts from the context's)
    fr->fcs = fcs;
    fr->has_checksum = (fhd >> 2) & 1;
    // ---- blocks (zstd.c:43075) and their section headers
    zl2_blk_rec* br = a.brec + (size_t)f * ZL2_MAXB;
    pos = hs;
    u32 nb = 0, lit_total = 0, seq_total = 0;
    for (;;) {
        if (n - pos < 3) { fr->err = ZL_E_SRCSIZE_WRONG; return; }
        const u32 bh = zl_rd_le(src + pos, 3);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
        pos += 3;
        if (type == 3) { fr->err = ZL_E_CORRUPTION; return; }
        const u32 csize = (type == 1) ? 1 : bsize;
        if (n - pos < csize) { fr->err = ZL_E_SRCSIZE_WRONG; return; }
        if (nb >= ZL2_MAXB) { fr->err = ZL_E_FALLBACK; atomicAdd(a.max_blocks + 1, 1u); return; }
        zl2_blk_rec r;
        r.type = type; r.size = bsize; r.src_off = pos;
        r.lit_mode = 0; r.lit_size = 0; r.lit_a = 0; r.lit_b = 0; r.seq_off = 0; r.seq_len = 0;
        r.lit_dst = lit_total; r.seq_dst = seq_total; r.nseq = 0; r.streams = 0; r.seq_modes = 0; r.pad[0] = 0; r.pad[1] = 0;
        if (type == 2) {
            u32 lit_bytes;
            const u32 perr = parse_block_sections(src + pos, bsize, r, lit_bytes);
            if (perr) { fr->err = perr; return; }
            lit_total += lit_bytes;
            seq_total += r.nseq + 2;  // + the totals and final-repcode records the sequence lanes append
        } else {
            if ((u64)bsize > fd.dst_cap) { fr->err = ZL_E_DSTSIZE_TOO_SMALL; return; }
        }
        br[nb++] = r;
        pos += csize;
        if (last) break;
    }
    // scratch capacity of this frame (planned on the host from dst_cap)
    if ((u64)lit_total > a.plan[f].lit_cap || (u64)seq_total > a.plan[f].seq_cap) { fr->err = ZL_E_CORRUPTION; return; }
    fr->nblocks = nb;
    fr->end_pos = pos;
    fr->cost = seq_total;
    fr->lit_cost = lit_total;
    atomicMax(a.max_blocks, nb);
    if (fr->has_checksum && n - pos < 4) { fr->err = ZL_E_CHECKSUM_WRONG; return; }
}

// ---------------------------------------------------------------------------------------
// zl2_order_kernel: one workgroup sorts the frames by decreasing sequence count (counting sort, 1024
// buckets).  The entropy kernels give a workgroup eight frames at a time and a workgroup lasts as
// long as its longest chain, so frames of similar length belong together, and the long ones first
// (measured on 10 000 reference-made 64 KiB frames of three kinds in turn: 12.9 -> ms below for the
// literal + sequence lanes).  Only the schedule depends on it: any permutation is correct.
// ---------------------------------------------------------------------------------------
#ifndef ZL2_SMALL
#define ZL2_SMALL 1  // 0: every frame through the lanes, as rounds 1-5 (zl2_small_kernel)
#endif
#ifndef ZL2_SPLIT_MIN
#define ZL2_SPLIT_MIN 1024   // frames from which on a batch may be cut in two (zl2_launch)
#endif
// One LDS add for all the lanes of a wave that name the same bucket (a batch of 100 000 small objects has a handful of costs: every lane of
// every wave used to queue on the same three words -- 287 us for config C4's batch; any permutation inside a bucket is correct).  Up to four
// buckets per wave this way, whoever is left adds alone.  Returns the lane's slot (the bucket's old count + the lane's rank among its peers).
__device__ __forceinline__ u32 order_bucket_add(u32* cnt, u32 bucket, bool on) {
    const u32 lane = threadIdx.x & 63;
    u64 todo = __ballot(on);
    u32 slot = 0;
    for (int r = 0; r < 4 && todo; r++) {
        const int leader = __builtin_ctzll(todo);
        const u32 lb = (u32)__shfl((int)bucket, leader, 64);
        const u64 same = __ballot(on && bucket == lb) & todo;
        u32 base = 0;
        if ((int)lane == leader) base = atomicAdd(&cnt[lb], (u32)__popcll(same));
        base = (u32)__shfl((int)base, leader, 64);
        if ((same >> lane) & 1) slot = base + (u32)__popcll(same & ((1ull << lane) - 1));
        todo &= ~same;
    }
    if ((todo >> lane) & 1) slot = atomicAdd(&cnt[bucket], 1u);
    return slot;
}

__global__ __launch_bounds__(1024) void zl2_order_kernel(zl2_args a) {
    __shared__ u32 cnt[1024];
    __shared__ u32 part[16];
    const u32 t = threadIdx.x, n = a.nframes;
    // Every frame already decoded by zl2_small_kernel (config C4: 100 000 small objects on a dictionary): nothing to sort, and the
    // entropy and exec launches behind this one return at once (zl2_part_range) -- they used to walk 100 000 finished frames, 0.6 ms.
    if (a.big == nullptr && a.max_blocks[3] == n) { if (t == 0) a.max_blocks[2] = ZL2_MODE_NONE; return; }
    for (u32 pass = 0; pass < 2; pass++) {  // 0: by sequences (order[0..n)), 1: by literal bytes (order[n..2n))
        const u32 shift = pass == 0 ? 5 : 7;
        cnt[t] = 0;
        __syncthreads();
        for (u32 f0 = 0; f0 < n; f0 += 1024) {  // (every wave stays whole: the ballots want all of its lanes)
            const u32 f = f0 + t;
            const u32 key = f < n ? (pass == 0 ? a.frec[f].cost : a.frec[f].lit_cost) >> shift : 0u;
            (void)order_bucket_add(cnt, 1023 - (key > 1023 ? 1023 : key), f < n);
        }
        __syncthreads();
        // exclusive scan over the 1024 buckets
        const u32 mine = cnt[t];
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 x = __shfl_up(inc, d, 64);
            if ((int)(t & 63) >= d) inc += x;
        }
        if ((t & 63) == 63) part[t >> 6] = inc;
        __syncthreads();
        u32 pre = 0;
        for (u32 w = 0; w < (t >> 6); w++) pre += part[w];
        __syncthreads();
        cnt[t] = pre + inc - mine;
        __syncthreads();
        if (pass == 0 && t == 0) {
            // How a large batch is scheduled (zl2_launch).  Single-block frames: in two halves side by side.  (A cut by cost --
            // frames below a half, a third, a quarter of the longest chain -- was measured on 10 000 reference-made 64 KiB
            // frames: the iid10 third has nearly as many sequences as the text third, the cut put both on the expensive side
            // and nothing moved; the even cut: 7.0 -> 6.3 ms.)  Multi-block frames: not cut (zl2_launch says what was tried).
            u32 gs = ZL2_MODE_WHOLE;
            if (n >= ZL2_SPLIT_MIN && *a.max_blocks <= 1) gs = ZL2_MODE_HALVES;
            a.max_blocks[2] = gs;
        }
        __syncthreads();
        for (u32 f0 = 0; f0 < n; f0 += 1024) {
            const u32 f = f0 + t;
            const u32 key = f < n ? (pass == 0 ? a.frec[f].cost : a.frec[f].lit_cost) >> shift : 0u;
            const u32 slot = order_bucket_add(cnt, 1023 - (key > 1023 ? 1023 : key), f < n);
            if (f < n) a.order[pass * n + slot] = f;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// FSE table description reader shared by the literal and sequence kernels
// (zstd.c:3269 FSE_readNCount_body).  Returns bytes consumed, 0 on error.
// ---------------------------------------------------------------------------------------
template <typename NormPtr>
__device__ __forceinline__ u32 read_ncount2(NormPtr norm, int* max_sym_io, int* log_out, int max_log, const u8* src, u32 n) {
    if (n < 1) return 0;
    i32 bit = 0;
    const i32 total = (i32)n * 8;
    int log = (int)zl_bits_bwd(src, n, bit, 4) + 5;
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
                const int r = (int)zl_bits_bwd(src, n, bit, 2);
                bit += 2;
                sym += r;
                if (r != 3) break;
            }
            if (sym > limit) return 0;
            prev_zero = false;
        }
        if (bit > total) return 0;
        const int max = (2 * threshold - 1) - remaining;
        const u32 peek = zl_bits_bwd(src, n, bit, (u32)nbits);
        int count;
        if ((int)(peek & (u32)(threshold - 1)) < max) { count = (int)(peek & (u32)(threshold - 1)); bit += nbits - 1; }
        else { count = (int)(peek & (u32)(2 * threshold - 1)); if (count >= threshold) count -= max; bit += nbits; }
        count--;
        remaining -= (count < 0) ? -count : count;
        norm[sym++] = (short)count;
        prev_zero = (count == 0);
        while (remaining < threshold && threshold > 1) { nbits--; threshold >>= 1; }
    }
    if (remaining != 1 || bit > total) return 0;
    *max_sym_io = sym - 1;
    *log_out = log;
    return (u32)((bit + 7) >> 3);
}

// ---------------------------------------------------------------------------------------
// zl2_lit_kernel: 64 lanes = 16 frames x 4 Huffman streams
// ---------------------------------------------------------------------------------------
#define LIT_FRAMES ZL2_WG_FRAMES
// MB = blocks per frame the instantiation has lanes for: ZL2_MAXB in general, 1 for launches in
// which no frame has more than one block (every frame a reference encoder makes from <= 128 KiB):
// those need an eighth of the ring space and of the threads, which leaves the CU's LDS to the
// sequence chains.  Which one runs is decided on the device (zl2_args::max_blocks, see zl2_launch).
template <int MB>
struct LitGroupT {
    static constexpr int LANES = MB * 4;  // stream lanes per frame
    u16 tab[2048];   // [7:0] symbol, [15:8] code length
    u32 ring[16 * LANES];  // 16 output dwords per stream lane, slot-major; the first 256 bytes hold the weights while a table is built
    u32 rin[32 * LANES];   // 32 input dwords per stream lane, slot-major: the compressed stream runs up to 31 dwords ahead of the decoder
    u32 wfse[64];    // weight FSE cells: [15:0] base, [23:16] nbits, [31:24] symbol
    short norm[16];
    u16 next[16];
    u8 cellsym[64];
    u32 rank_start[16];
    u32 log;
    u32 ok;          // table usable by the blocks that follow it
    u32 tree;        // size of the tree description of the block that brought the table
    u32 pad;
};
static_assert(sizeof(LitGroupT<ZL2_MAXB>) % 16 == 0 && sizeof(LitGroupT<1>) % 16 == 0, "LitGroup must stay 16-byte aligned");
static_assert(sizeof(u32) * 16 * LitGroupT<1>::LANES >= 256, "the ring region doubles as weights[256]");

// Huffman tree description -> g.tab (zstd.c:3470 HUF_readStats_body, zstd.c:38436
// HUF_readDTableX1_wksp).  One lane.  Returns bytes consumed, 0 on error, 0xFFFFFFFF when the
// table needs 12 bits (handled by the general kernel).
template <class Group>
__device__ __forceinline__ u32 build_huf_table(Group& g, const u8* src, u32 n) {
    u8* const weights = (u8*)g.ring;
    if (n < 1) return 0;
    int nsym = 0;
    u32 used;
    const u32 hb = src[0];
    if (hb >= 128) {
        nsym = (int)hb - 127;
        used = 1 + (u32)(nsym + 1) / 2;
        if (used > n) return 0;
        for (int i = 0; i < nsym; i++) {
            const u8 b = src[1 + i / 2];
            weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
    } else {
        used = 1 + hb;
        if (used > n || hb < 2) return 0;
        int max_sym = 12, log = 0;  // weights are 0..12
        const u32 hs = read_ncount2(g.norm, &max_sym, &log, 6, src + 1, hb);
        if (hs == 0 || hs >= hb) return 0;
        // weight FSE table (zstd.c:3690)
        const int size = 1 << log;
        int high = size - 1;
        for (int s = 0; s <= max_sym; s++) {
            if (g.norm[s] == -1) { g.cellsym[high--] = (u8)s; g.next[s] = 1; }
            else g.next[s] = (u16)g.norm[s];
        }
        const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
        int pos = 0;
        for (int s = 0; s <= max_sym; s++)
            for (int i = 0; i < g.norm[s]; i++) {
                g.cellsym[pos] = (u8)s;
                do { pos = (pos + step) & mask; } while (pos > high);
            }
        if (pos != 0) return 0;
        for (int u = 0; u < size; u++) {
            const u32 s = g.cellsym[u];
            const u32 ns = g.next[s]++;
            const u32 nb = (u32)log - (31 - __clz(ns));
            g.wfse[u] = (((ns << nb) - (u32)size) & 0xFFFF) | (nb << 16) | (s << 24);
        }
        const u8* bs = src + 1 + hs;
        const u32 bn = hb - hs;
        i32 bp = zl_bwd_init(bs, bn);
        if (bp < 0) return 0;
        bp -= log;
        u32 s1 = zl_bits_bwd(bs, bn, bp, (u32)log);
        bp -= log;
        u32 s2 = zl_bits_bwd(bs, bn, bp, (u32)log);
        if (bp < 0) return 0;
        for (;;) {
            if (nsym > 253) return 0;
            u32 e = g.wfse[s1];
            weights[nsym++] = (u8)(e >> 24);
            u32 nb = (e >> 16) & 0xFF;
            bp -= (i32)nb;
            s1 = (e & 0xFFFF) + zl_bits_bwd(bs, bn, bp, nb);
            if (bp < 0) { weights[nsym++] = (u8)(g.wfse[s2] >> 24); break; }
            if (nsym > 253) return 0;
            e = g.wfse[s2];
            weights[nsym++] = (u8)(e >> 24);
            nb = (e >> 16) & 0xFF;
            bp -= (i32)nb;
            s2 = (e & 0xFFFF) + zl_bits_bwd(bs, bn, bp, nb);
            if (bp < 0) { weights[nsym++] = (u8)(g.wfse[s1] >> 24); break; }
        }
    }
    u32 total = 0;
    for (int w = 0; w < 16; w++) g.rank_start[w] = 0;
    for (int i = 0; i < nsym; i++) {
        const u32 w = weights[i];
        if (w > ZL_HUF_LOG_MAX) return 0;
        g.rank_start[w]++;
        if (w) total += 1u << (w - 1);
    }
    if (total == 0) return 0;
    const u32 log = 32 - __clz(total);
    if (log > ZL_HUF_LOG_MAX) return 0;
    const u32 rest = (1u << log) - total;
    if (rest & (rest - 1)) return 0;
    const u32 last_w = 32 - __clz(rest);
    weights[nsym++] = (u8)last_w;
    g.rank_start[last_w]++;
    if (g.rank_start[1] < 2 || (g.rank_start[1] & 1)) return 0;  // zstd.c:3549
    if (log > 11) return 0xFFFFFFFFu;
    // counts -> first cell of each weight class
    u32 cell = 0;
    for (u32 w = 1; w <= log; w++) {
        const u32 c = g.rank_start[w];
        g.rank_start[w] = cell;
        cell += c << (w - 1);
    }
    for (int s = 0; s < nsym; s++) {
        const u32 w = weights[s];
        if (!w) continue;
        const u32 cnt = 1u << (w - 1);
        const u32 start = g.rank_start[w];
        g.rank_start[w] = start + cnt;
        const u16 e = (u16)((u32)s | ((log + 1 - w) << 8));
        for (u32 k = 0; k < cnt; k++) g.tab[start + k] = e;
    }
    g.log = log;
    return used;
}

// one backward Huffman stream on one lane (zstd.c:38560).  Main loop: left-aligned accumulator,
// two symbols per refill check, four symbols per dword store, no branches; the last few symbols
// (and short streams) go through the generic reader so that the end-of-stream test stays exact.
template <int LIT_LANES>
__device__ __forceinline__ bool huf_stream2(const u16* tab, u32 log, u8* dst, u32 nout, const u8* src, u32 n, u32* ring /* this lane's 16 output dwords in LDS, LIT_LANES apart */,
                                            u32* rin /* this lane's 16 input dwords in LDS, LIT_LANES apart */) {
    i32 pos = zl_bwd_init(src, n);
    if (pos < 0) return false;
    u32 i = 0;
    const u32 head = (u32)((4 - ((uintptr_t)dst & 3)) & 3);
    for (; i < head && i < nout; i++) {
        const u32 e = tab[zl_bits_bwd(src, n, pos - (i32)log, log)];
        dst[i] = (u8)e;
        pos -= (i32)(e >> 8);
    }
    if (pos >= 96 && log >= 1) {
        zl_lreader r;
        zl_lr_open(r, src, n, pos);
        const u32 sh = 32 - log;
        const u32* w = r.w;
        u64 acc = r.acc;
        i32 nav = r.nav, idx = r.idx;
        const i32 nav_start = nav, idx_start = idx;
        // Memory stays off the decode chain on both sides.  The compressed dwords reach the lane through
        // a 16-entry LDS ring that is refilled eight dwords at a time, two iterations after the loads
        // were issued (a dword loaded when it is needed costs a memory round trip per four symbols:
        // measured 2800 cycles per iteration against ~750 for the table lookups).  Decoded dwords leave
        // through a second ring, 64 bytes at a time: loads and stores share vmcnt and complete out of
        // order, so only vmcnt(0) is safe, and it also waits for every store in flight.
        if (i + 4 <= nout && idx >= 3) {
            i32 low = idx >= 15 ? idx - 15 : 0;  // lowest dword present in the ring
            if (idx >= 15) {
                uint4 q[4];
#pragma unroll
                for (int k = 0; k < 4; k++) __builtin_memcpy(&q[k], w + low + 4 * k, 16);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    rin[((low + 4 * k + 0) & 31) * LIT_LANES] = q[k].x;
                    rin[((low + 4 * k + 1) & 31) * LIT_LANES] = q[k].y;
                    rin[((low + 4 * k + 2) & 31) * LIT_LANES] = q[k].z;
                    rin[((low + 4 * k + 3) & 31) * LIT_LANES] = q[k].w;
                }
            } else {
                for (i32 k = 0; k <= idx; k++) rin[(k & 31) * LIT_LANES] = w[k];
            }
            uint4 pv0 = make_uint4(0, 0, 0, 0), pv1 = pv0;  // eight dwords on their way from memory
            u32 pend = 0;                                    // iterations until they are written into the ring
            u32 it = 0;
            u8* const out0 = dst + i;
            while (i + 4 <= nout && idx >= 3) {
                const u32 wa = rin[((u32)idx & 31) * LIT_LANES], wb = rin[((u32)(idx - 1) & 31) * LIT_LANES];
                const u32 e0 = tab[(u32)(acc >> 32) >> sh];
                acc <<= (e0 >> 8);
                const u32 e1 = tab[(u32)(acc >> 32) >> sh];
                acc <<= (e1 >> 8);
                nav -= (i32)((e0 >> 8) + (e1 >> 8));
                const bool f0 = nav <= 32;
                acc |= f0 ? ((u64)wa << (32 - nav)) : 0ull;
                nav += f0 ? 32 : 0;
                const u32 e2 = tab[(u32)(acc >> 32) >> sh];
                acc <<= (e2 >> 8);
                const u32 e3 = tab[(u32)(acc >> 32) >> sh];
                acc <<= (e3 >> 8);
                nav -= (i32)((e2 >> 8) + (e3 >> 8));
                const bool f1 = nav <= 32;
                const u32 wsel = f0 ? wb : wa;
                acc |= f1 ? ((u64)wsel << (32 - nav)) : 0ull;
                nav += f1 ? 32 : 0;
                idx -= (f0 ? 1 : 0) + (f1 ? 1 : 0);
                ring[(it & 15) * LIT_LANES] = (e0 & 0xFF) | ((e1 & 0xFF) << 8) | ((e2 & 0xFF) << 16) | (e3 << 24);
                if (pend && --pend == 0) {  // the dwords asked for two iterations ago go below the ones still unread
                    low -= 8;
                    rin[((u32)(low + 0) & 31) * LIT_LANES] = pv0.x;
                    rin[((u32)(low + 1) & 31) * LIT_LANES] = pv0.y;
                    rin[((u32)(low + 2) & 31) * LIT_LANES] = pv0.z;
                    rin[((u32)(low + 3) & 31) * LIT_LANES] = pv0.w;
                    rin[((u32)(low + 4) & 31) * LIT_LANES] = pv1.x;
                    rin[((u32)(low + 5) & 31) * LIT_LANES] = pv1.y;
                    rin[((u32)(low + 6) & 31) * LIT_LANES] = pv1.z;
                    rin[((u32)(low + 7) & 31) * LIT_LANES] = pv1.w;
                }
                // Fewer than eight unread dwords: ask for eight more -- and with this lane every lane of the wave
                // that has room for them (the ring holds 32), so that the lanes refill in the same iterations:
                // decided lane by lane, some lane of the wave was refilling in nearly every iteration and the whole
                // wave walked through the refill code each time.
                const bool want = !pend && low > 0 && idx - low < 8;
                if (__ballot(want) != 0 && !pend && low > 0 && idx - low < 24) {
                    if (low >= 8) {
                        __builtin_memcpy(&pv0, w + low - 8, 16);
                        __builtin_memcpy(&pv1, w + low - 4, 16);
                        pend = 2;
                    } else {  // the last few dwords of the stream
                        for (i32 k = 0; k < low; k++) rin[((u32)k & 31) * LIT_LANES] = w[k];
                        low = 0;
                    }
                }
                if ((it & 15) == 15) {
                    u32* o = (u32*)(out0 + (size_t)(it - 15) * 4);
#pragma unroll
                    for (int q = 0; q < 16; q++) o[q] = ring[q * LIT_LANES];
                }
                it++;
                i += 4;
            }
            for (u32 k = it & ~15u; k < it; k++) *(u32*)(out0 + (size_t)k * 4) = ring[(k & 15) * LIT_LANES];
        }
        // exact bit position: every consumed bit left the accumulator or was replaced by a refill
        pos = r.pos - ((nav_start - nav) + 32 * (idx_start - idx));
    }
    for (; i < nout; i++) {
        const u32 e = tab[zl_bits_bwd(src, n, pos - (i32)log, log)];
        dst[i] = (u8)e;
        pos -= (i32)(e >> 8);
    }
    return pos == 0;  // zstd.c:38640 BIT_endOfDStream
}

// One lane per (frame, block, stream).  A block that brings a tree is the "leader" of the treeless
// blocks behind it; leaders are processed in block order (round r builds block r's table, then every
// lane whose leader is r decodes), so frames whose blocks share one table -- everything the HIP
// encoder writes -- finish in one round with all lanes busy.
template <int MB>
__device__ __forceinline__ void lit_role(const zl2_args& a, LitGroupT<MB>* G, u32* rounds, u32 wg) {
    const u32 t = threadIdx.x;
    const u32 sub = t & 3, grp = (t >> 2) % LIT_FRAMES, b = t / (4 * LIT_FRAMES);
    const u32 slot = wg * LIT_FRAMES + grp;
    // longest literal streams first -- unless the batch is cut in two: both roles must then agree on which frames a part holds
    u32 pg0, pg1; bool psplit;
    zl2_part_range(a, pg0, pg1, psplit);
    const u32 f = slot < a.nframes ? a.order[(psplit ? 0u : a.nframes) + slot] : 0xFFFFFFFFu;
    LitGroupT<MB>& g = G[grp];
    if (t < LIT_FRAMES) G[t].ok = 0;
    if (t == 0) *rounds = 0;
    __syncthreads();
    bool active = false;
    int lb = -1;
    zl2_blk_rec r;
    r.type = 0;
    const u8* src = nullptr;
    u8* lit = nullptr;
    zl2_frame_rec* fr = nullptr;
    // large-frame pipeline: the tree a segment inherits from an earlier block of the frame is built in a round of its own
    // (round index 0; block k of the segment leads round k + 1)
    bool ext_leader = false;
    if (f < a.nframes) {
        fr = a.frec + f;
        const u32 nblocks = (fr->err == 0 && !fr->skippable) ? fr->nblocks : 0;
        const bool have_ext = a.ext != nullptr && nblocks != 0 && ext_rec(a, f, 0).type == 2;
        ext_leader = have_ext && b == 0 && sub == 0;
        if (b < nblocks) {
            const zl2_blk_rec* br = a.brec + (size_t)f * ZL2_MAXB;
            src = a.src + a.frames[f].src_off;
            lit = a.lit_scratch + a.plan[f].lit_off;
            for (u32 k = 0; k <= b; k++)
                if (br[k].type == 2 && br[k].lit_mode == 2) lb = (int)k;
            r = br[b];
            active = (r.type == 2 && r.lit_mode >= 2);
            if (active && lb < 0 && !have_ext) { set_err(fr, ZL_E_DICT_CORRUPTED); active = false; }  // treeless block without a previous table (zstd.c:43171)
        } else if (ext_leader) {
            src = a.src + a.frames[f].src_off;
        }
    }
    if (active) atomicOr(rounds, 1u << (lb + 1));
    __syncthreads();
    const u32 mask = *rounds;
    for (u32 ri = 0; ri <= MB; ri++) {
        if (!((mask >> ri) & 1)) continue;
        const int round = (int)ri - 1;
        const bool mine = active && lb == round;
        if (ri == 0 ? ext_leader : (mine && (int)b == round && sub == 0)) {
            zl2_blk_rec lr = r;
            if (ri == 0) lr = ext_rec(a, f, 0);
            const u8* blk = (ri == 0 ? ext_src(a, f, lr) : src) + lr.src_off;
            u32 tree;
            if (ri == 0 && a.dict_pre != nullptr) {  // the dictionary's table, built once per context: copied
                const zl2_dict_pre* P = a.dict_pre;
                tree = P->huf_ok == 1 ? lr.lit_b : (P->huf_ok == 2 ? 0xFFFFFFFFu : 0u);
                if (P->huf_ok == 1) {
                    const uint4* from = (const uint4*)P->tab;
                    uint4* to = (uint4*)g.tab;
                    for (u32 i = 0; i < 2048 * 2 / 16; i += 8) {
                        uint4 v[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) v[q] = from[i + q];
#pragma unroll
                        for (int q = 0; q < 8; q++) to[i + q] = v[q];
                    }
                    g.log = P->huf_log;
                }
            } else tree = build_huf_table(g, blk + lr.lit_a, lr.lit_b);
            if (tree == 0xFFFFFFFFu) { set_err(fr, ZL_E_FALLBACK); atomicAdd(a.max_blocks + 1, 1u); tree = 0; g.ok = 0; }
            else if (tree == 0) { set_err(fr, ZL_E_CORRUPTION); g.ok = 0; }
            else g.ok = 1;
            g.tree = tree;
        }
        __syncthreads();
        if (mine && g.ok && fr->err == 0) {
            const u32 tree = ((int)b == round) ? g.tree : 0u;
            const u8* blk = src + r.src_off;
            const u8* hs = blk + r.lit_a + tree;
            const u32 cs = r.lit_b - tree, regen = r.lit_size;
            u8* out = lit + r.lit_dst;
            u32* ring = g.ring + (b * 4 + sub);
            u32* rin = g.rin + (b * 4 + sub);
            bool good = true;
            if (r.streams == 1) {
                if (sub == 0) good = huf_stream2<MB * 4>(g.tab, g.log, out, regen, hs, cs, ring, rin);
            } else if (cs < 10) {
                good = false;
            } else {
                const u32 s1 = zl_rd_le(hs, 2), s2 = zl_rd_le(hs + 2, 2), s3 = zl_rd_le(hs + 4, 2);
                const u32 seg = (regen + 3) / 4;
                if (6 + s1 + s2 + s3 > cs || seg * 3 > regen) good = false;
                else {
                    const u32 s4 = cs - 6 - s1 - s2 - s3;
                    const u32 so = sub == 0 ? 0 : (sub == 1 ? s1 : (sub == 2 ? s1 + s2 : s1 + s2 + s3));
                    const u32 sl = sub == 0 ? s1 : (sub == 1 ? s2 : (sub == 2 ? s3 : s4));
                    const u32 cnt = (sub == 3) ? regen - 3 * seg : seg;
                    good = huf_stream2<MB * 4>(g.tab, g.log, out + sub * seg, cnt, hs + 6 + so, sl, ring, rin);
                }
            }
            if (!good) set_err(fr, ZL_E_CORRUPTION);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// zl2_seq_kernel: one lane per frame, 16 frames per 64-lane workgroup
// ---------------------------------------------------------------------------------------
#define SEQ_FRAMES ZL2_WG_FRAMES
struct __align__(16) SeqLane {
    // one entry per FSE state, LL [0,512) ML [512,1024) OF [1024,1280):
    // [9:0] next-state base, [13:10] state bits, [21:16] symbol, [28:24] extra bits of the symbol
    u32 ent[1280];
};
// Table construction scratch of one frame; lives in the record ring, which is idle while tables are built.
struct SeqBuild {
    short norm[3][64];  // normalised counts per table (kind 0 LL, 1 OF, 2 ML)
    u32 next[3][32];    // two 16-bit counters per word
    u8 hdr[192];        // the table descriptions, copied out of the frame once
    u32 mode[3], log[3], max_sym[3], rle[3];
    u32 stamp;          // round + 1 of the block whose descriptions these are
    u32 pad[3];
};
static_assert(sizeof(SeqBuild) <= 1024, "one KiB of ring per frame");
struct SeqShared {
    SeqLane L[SEQ_FRAMES];     // one table set per frame, shared by the lanes of its blocks
    uint4 ring[8][64];         // 8 finished records per lane, slot-major (conflict-free)
    u32 logs[SEQ_FRAMES][4];   // table logs LL, OF, ML of the frame's current tables; [3]: 0 none yet, 1 valid, 2 broken
    u32 ll_tab[36];  // [23:0] baseline, [31:24] extra bits
    u32 ml_tab[53];
    short defn[3][64];  // predefined distributions (kind 0 LL, 1 OF, 2 ML), zstd.c:15480-15512
    u32 val_tab[3][64];  // quad chains: baseline of symbol s, one table per role (0 LL, 1 OF: 1 << s, 2 ML)
    u32 qpos[SEQ_FRAMES];  // quad chains: where the frame's bitstream starts this round (0xFFFFFFFF: no chain to run)
    u32 cmd;               // two-wave chains: 1 = a chain phase follows the barrier, 0 = the role is over (wave 0 tells wave 1)
};
// Repcode history of a block that is decoded before its predecessor has finished: the three
// unknown start values travel as symbols, 0x80000000 | k + 4 * (times "minus one" was applied)
// (zstd.c:44291-44327 only copies history entries around and, for offset code 3 with no literals,
// subtracts one).  zl2_exec_kernel, which walks the blocks in order, puts the numbers in.
#define ZL2_REP_SYM 0x80000000u
__device__ __forceinline__ u32 rep_minus1(u32 r) { return (r >> 31) ? r + 4 : r - 1; }

// zstd.c:43497-43600 ZSTD_buildFSETable_body, compact cells.  One lane.
__device__ __forceinline__ u32 ll_extra_bits(u32 s) {
    return s < 16 ? 0u : (s >= 25 ? s - 19 : (0x46d2249u >> (3 * (s - 16))) & 7u);
}
__device__ __forceinline__ u32 ml_extra_bits(u32 s) {
    return s < 32 ? 0u : (s >= 43 ? s - 36 : (u32)(0x1646d2249ull >> (3 * (s - 32))) & 7u);
}

// symbol and its extra-bit count in entry format; base 0: literal lengths, 512: match lengths, 1024: offsets
__device__ __forceinline__ u32 seq_sym_bits(u32 base, u32 s) {
    const u32 x = base == 0 ? ll_extra_bits(s) : (base == 512 ? ml_extra_bits(s) : s);
    return (s << 16) | (x << 24);
}
// FSE decoding table from normalised counts (zstd.c:43497 ZSTD_buildFSETable_body).  `next` holds the
// running "next state" counter of two symbols per word; the cell loop takes its numbers with
// returning LDS atomics, eight at a time: a plain read-increment-write through LDS would make every
// cell wait for the previous one (two LDS round trips per cell), the atomics only queue up.
__device__ __forceinline__ bool build_seq_table(SeqLane& L, u32 base, const short* norm, u32* next /* 32 words */, int max_sym, int log) {
    const int size = 1 << log;
    int high = size - 1;
    for (int s = 0; s <= max_sym; s += 2) {
        const int n0 = norm[s], n1 = (s + 1 <= max_sym) ? norm[s + 1] : 0;
        if (n0 == -1) L.ent[base + high--] = (u32)s;
        if (n1 == -1) L.ent[base + high--] = (u32)(s + 1);
        next[s >> 1] = (u32)(n0 == -1 ? 1 : n0) | ((u32)(n1 == -1 ? 1 : n1) << 16);
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s <= max_sym; s++) {
        const int n = norm[s];  // (a reload per cell would put an LDS round trip into the loop: the stores may alias)
        for (int i = 0; i < n; i++) {
            L.ent[base + pos] = (u32)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (int u0 = 0; u0 < size; u0 += 8) {  // size is a power of two >= 32 (table logs are >= 5)
        u32 sy[8], ns[8];
#pragma unroll
        for (int q = 0; q < 8; q++) sy[q] = L.ent[base + u0 + q];
#pragma unroll
        for (int q = 0; q < 8; q++) ns[q] = (atomicAdd(&next[sy[q] >> 1], 1u << (16 * (sy[q] & 1))) >> (16 * (sy[q] & 1))) & 0xFFFF;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const u32 nb = (u32)log - (31 - __clz(ns[q]));
            L.ent[base + u0 + q] = (((ns[q] << nb) - (u32)size) & 0x3FF) | (nb << 10) | seq_sym_bits(base, sy[q]);
        }
    }
    return true;
}

// kind: 0 LL, 1 OF, 2 ML.  Parses one table description (zstd.c:43640 ZSTD_buildSeqTable) out of the
// header copy; returns bytes consumed or 0xFFFFFFFF.
__device__ __forceinline__ u32 parse_seq_table(SeqBuild& B, int kind, u32 mode, const u8* src, u32 n, bool have_prev) {
    const int max_sym = kind == 0 ? ZL_MAX_LL : (kind == 2 ? ZL_MAX_ML : ZL_MAX_OF);
    const int max_log = kind == 1 ? ZL_OF_LOG_MAX : ZL_LL_LOG_MAX;
    B.mode[kind] = mode;
    if (mode == 0) return 0;
    if (mode == 1) {
        if (n < 1 || (int)src[0] > max_sym) return 0xFFFFFFFFu;
        B.rle[kind] = src[0];
        return 1;
    }
    if (mode == 2) {
        int ms = max_sym, log = 0;
        const u32 used = read_ncount2(B.norm[kind], &ms, &log, max_log, src, n);
        if (used == 0) return 0xFFFFFFFFu;
        B.max_sym[kind] = (u32)ms;
        B.log[kind] = (u32)log;
        return used;
    }
    return have_prev ? 0 : 0xFFFFFFFFu;
}
// builds table `kind` of the frame from its parsed description; returns the table log or 0xFFFFFFFF
__device__ __forceinline__ u32 make_seq_table(SeqLane& L, SeqBuild& B, const short* defn, int kind, u32 old_log) {
    const u32 base = kind == 0 ? 0u : (kind == 2 ? 512u : 1024u);
    const u32 mode = B.mode[kind];
    if (mode == 1) { L.ent[base] = seq_sym_bits(base, B.rle[kind]); return 0; }
    if (mode == 3) return old_log;
    // predefined and described tables go through ONE call of the builder: lanes that build different
    // kinds of table stay converged
    int max_sym = (int)B.max_sym[kind], log = (int)B.log[kind];
    const short* norm = B.norm[kind];
    if (mode == 0) {
        max_sym = kind == 0 ? 35 : (kind == 2 ? 52 : 28);
        log = kind == 1 ? 5 : 6;
        norm = defn + 64 * kind;  // the predefined distributions, staged in LDS once per workgroup
    }
    return build_seq_table(L, base, norm, B.next[kind], max_sym, log) ? (u32)log : 0xFFFFFFFFu;
}

// The chain from sequence i on, one field at a time with every bound checked: the end of every chain, and the
// whole chain of short streams.  Writes the block's totals and final history behind the records.
__device__ __forceinline__ u32 seq_tail(SeqShared& S, const SeqLane& L, const zl2_blk_rec& r, const u8* bs, u32 bn, i32 bp, u32 i, u32 sl, u32 so, u32 sm,
                                        u32 r0, u32 r1, u32 r2, u32 lit_used, u32 out_used, uint4* oseq) {
    const u32 nseq = r.nseq;
    // generic tail (also the whole chain of short streams)
    for (; i < nseq; i++) {
        const u32 cl = L.ent[sl], cm = L.ent[512 + sm], co = L.ent[1024 + so];
        const u32 yl = (cl >> 16) & 0xFF, ym = (cm >> 16) & 0xFF, yo = (co >> 16) & 0xFF;
        if (yl > ZL_MAX_LL || ym > ZL_MAX_ML || yo > ZL_MAX_OF) return ZL_E_CORRUPTION;
        const u32 tl = S.ll_tab[yl], tm = S.ml_tab[ym];
        const u32 nl = tl >> 24, nm = tm >> 24;
        bp -= (i32)yo;
        const u32 ofv = (1u << yo) + zl_bits_bwd(bs, bn, bp, yo);
        bp -= (i32)nm;
        const u32 ml = (tm & 0xFFFFFF) + zl_bits_bwd(bs, bn, bp, nm);
        bp -= (i32)nl;
        const u32 ll = (tl & 0xFFFFFF) + zl_bits_bwd(bs, bn, bp, nl);
        if (bp < 0) return ZL_E_CORRUPTION;
        u32 off;
        if (ofv > 3) { off = ofv - 3; r2 = r1; r1 = r0; r0 = off; }
        else {
            const u32 idx = ofv - 1 + (ll == 0);
            if (idx == 0) off = r0;
            else {
                off = (idx == 3) ? rep_minus1(r0) : (idx == 1 ? r1 : r2);
                if (off == 0) return ZL_E_CORRUPTION;
                if (idx != 1) r2 = r1;
                r1 = r0;
                r0 = off;
            }
        }
        oseq[i] = make_uint4(lit_used, out_used, off, 0u);
        lit_used += ll;
        out_used += ll + ml;
        if (out_used > ZL_BLOCK_MAX) return ZL_E_CORRUPTION;
        if (i + 1 < nseq) {
            u32 nb = (cl >> 10) & 0xF;
            bp -= (i32)nb;
            sl = (cl & 0x3FF) + zl_bits_bwd(bs, bn, bp, nb);
            nb = (cm >> 10) & 0xF;
            bp -= (i32)nb;
            sm = (cm & 0x3FF) + zl_bits_bwd(bs, bn, bp, nb);
            nb = (co >> 10) & 0xF;
            bp -= (i32)nb;
            so = (co & 0x3FF) + zl_bits_bwd(bs, bn, bp, nb);
            if (bp < 0) return ZL_E_CORRUPTION;
        }
    }
    if (bp != 0) return ZL_E_CORRUPTION;  // zstd.c:44690
    if (lit_used > r.lit_size) return ZL_E_CORRUPTION;
    oseq[nseq] = make_uint4(lit_used, out_used, 1u, 0u);  // totals of the block
    oseq[nseq + 1] = make_uint4(r0, r1, r2, 0u);          // history handed to the next block
    return 0;
}

// The 3-state chain of one block on one lane (zstd.c:44241-44356).  Returns 0 or an error code.
__device__ __forceinline__ u32 seq_chain(SeqShared& S, const SeqLane& L, u32 lane, const zl2_blk_rec& r, const u8* sp, u32 pos,
                                         u32 ll_log, u32 of_log, u32 ml_log, bool symbolic, uint4* oseq) {
    u32 r0 = symbolic ? ZL2_REP_SYM : 1u, r1 = symbolic ? (ZL2_REP_SYM | 1u) : 4u, r2 = symbolic ? (ZL2_REP_SYM | 2u) : 8u;
    // ---- the 3-state chain (zstd.c:44241-44356)
    const u8* bs = sp + pos;
    const u32 bn = r.seq_len - pos;
    i32 bp = zl_bwd_init(bs, bn);
    if (bp < 0) return ZL_E_CORRUPTION;
    const u32 nseq = r.nseq;
    u32 lit_used = 0, out_used = 0;  // running literal / output positions: what the exec kernel consumes
    u32 i = 0;
    u32 sl, so, sm;
    bool fast = bp >= 128;
    zl_lreader rd;
    if (fast) {
        zl_lr_open(rd, bs, bn, bp);
        sl = zl_lr_take(rd, ll_log);
        if (rd.nav <= 32) zl_lr_refill(rd);
        so = zl_lr_take(rd, of_log);
        sm = zl_lr_take(rd, ml_log);
        if (rd.nav <= 32) zl_lr_refill(rd);
        // Fast loop.  The dwords a sequence may need (w[idx-3..idx]) are loaded one sequence ahead;
        // fields are cut group by group (offset bits | length bits | state bits), one refill check
        // after each.  (Cutting all three from one window when they fit 32 bits was measured: the
        // extra divergent path made the loop 9 % slower.)
        const u32* w = rd.w;
        u64 acc = rd.acc;
        i32 nav = rd.nav, idx = rd.idx;
        const i32 nav_start = nav, idx_start = idx;
        u32 bad = 0;
        typedef u32 zl_v4u __attribute__((ext_vector_type(4)));
        zl_v4u t4 = {0, 0, 0, 0};
        const bool go = nseq > 0 && idx >= 5;
        if (go) t4 = *(const zl_v4u*)(w + idx - 3);
        // Records go to a 16-entry LDS ring and are
        // written out 128 bytes at a time.  Loads and stores share vmcnt and may complete out of
        // order, so the wait for the refill dwords is vmcnt(0), which also waits for every store
        // in flight: one store per sequence cost ~1 ms per 12k sequences (measured 9.6 -> 8.6 ms).
        zl_v4u* ring = (zl_v4u*)&S.ring[0][lane];
#define ZL_TAKE(dst_, n_)                                              \
        {                                                            \
            const u32 n__ = (n_);                                    \
            dst_ = n__ ? ((u32)(acc >> 32) >> (32 - n__)) : 0u;      \
            acc <<= n__;                                             \
            nav -= (i32)n__;                                         \
            const bool f__ = nav <= 32;                              \
            const u32 ws__ = used == 0 ? t4.w : (used == 1 ? t4.z : t4.y); \
            acc |= f__ ? ((u64)ws__ << (32 - nav)) : 0ull;           \
            nav += f__ ? 32 : 0;                                     \
            used += f__ ? 1 : 0;                                     \
        }
        // The three table entries of the NEXT sequence are asked for as soon as its states are known -- in front of the
        // values, the repcode rules and the record of this one -- and carried round the loop, and the stream's next dwords
        // right behind them.  (volatile, through LDS pointers: left to itself the compiler sinks the whole state
        // computation behind the record store, because nothing in this iteration uses it, and the loop's dependent chain
        // -- look-up, field cuts, states, look-up -- then runs end to end through everything else.)
        typedef __attribute__((address_space(3))) const volatile u32 lds_cvu32;
        const u32 ent_at = (u32)(uintptr_t)&L.ent[0];
        u32 cl = 0, cm = 0, co = 0;
        if (go) { cl = *(lds_cvu32*)(uintptr_t)(ent_at + 4 * sl); cm = *(lds_cvu32*)(uintptr_t)(ent_at + 2048 + 4 * sm); co = *(lds_cvu32*)(uintptr_t)(ent_at + 4096 + 4 * so); }
        for (; go && i < nseq && idx >= 5; i++) {
            u32 used = 0;
            const u32 yl = (cl >> 16) & 0xFF, ym = (cm >> 16) & 0xFF, yo = co >> 24;
            const u32 nl = cl >> 24, nm = cm >> 24;
            const u32 bl = (cl >> 10) & 0xF, bm = (cm >> 10) & 0xF, bo = (co >> 10) & 0xF;
            const u32 n2 = nm + nl;
            const u32 sb = (i + 1 < nseq) ? bl + bm + bo : 0u;
            u32 ofx, x, u;
            {
                ZL_TAKE(ofx, yo);
                ZL_TAKE(x, n2);
                ZL_TAKE(u, sb);
            }
            // next states first: they are the loop-carried chain (LL bits on top, then ML, then OF)
            so = (co & 0x3FF) + (u & ((1u << bo) - 1));
            sm = (cm & 0x3FF) + ((u >> bo) & ((1u << bm) - 1));
            sl = (cl & 0x3FF) + (u >> (bo + bm));
            cl = *(lds_cvu32*)(uintptr_t)(ent_at + 4 * sl);
            cm = *(lds_cvu32*)(uintptr_t)(ent_at + 2048 + 4 * sm);
            co = *(lds_cvu32*)(uintptr_t)(ent_at + 4096 + 4 * so);
            idx -= (i32)used;
#ifndef ZL_EXP_NOLOAD
            t4 = *(const zl_v4u*)(w + idx - 3);
#endif
            const u32 tl = S.ll_tab[yl], tm = S.ml_tab[ym];  // baselines: off the critical chain
            const u32 ofv = (1u << yo) + ofx;
            const u32 ml = (tm & 0xFFFFFF) + (x >> nl);  // match-length bits above literal-length bits
            const u32 ll = (tl & 0xFFFFFF) + (x & ((1u << nl) - 1));
            // repcode rules (zstd.c:44291-44327) as mask arithmetic: written with ?: the compiler builds exec-masked branches
            const u32 m_real = 0u - (u32)(ofv > 3);
            const u32 ri = ofv - 1 + (u32)(ll == 0);  // meaningful when not real: 0..3
            const u32 m1 = 0u - (u32)(ri == 1), m2 = 0u - (u32)(ri == 2), m3 = 0u - (u32)(ri == 3);
            const u32 r0m = r0 + 5 * (r0 >> 31) - 1;  // rep_minus1
            const u32 cand = (r1 & m1) | (r2 & m2) | (r0m & m3) | (r0 & ~(m1 | m2 | m3));
            const u32 off = ((ofv - 3) & m_real) | (cand & ~m_real);
            const u32 m_sh2 = m_real | m2 | m3, m_sh1 = m_sh2 | m1;
            r2 = (r1 & m_sh2) | (r2 & ~m_sh2);
            r1 = (r0 & m_sh1) | (r1 & ~m_sh1);
            r0 = off;
            bad |= (u32)(off == 0);
            zl_v4u rec = {lit_used, out_used, off, 0u};
            lit_used += ll;
            out_used += ll + ml;
            bad |= (out_used > ZL_BLOCK_MAX) ? 1u : 0u;
            ring[(i & 7) * 64] = rec;
            if ((i & 7) == 7) {
                zl_v4u* o = (zl_v4u*)(oseq + (i - 7));
#pragma unroll
                for (int q = 0; q < 8; q++) o[q] = ring[q * 64];
            }
        }
        for (u32 k = i & ~7u; k < i; k++) { const zl_v4u v = ring[(k & 7) * 64]; oseq[k] = make_uint4(v.x, v.y, v.z, v.w); }
#undef ZL_TAKE
        if (bad) return ZL_E_CORRUPTION;
        // exact bit position: every consumed bit left the accumulator or was replaced by a refill
        rd.pos -= (nav_start - nav) + 32 * (idx_start - idx);
        // hand the exact bit position to the generic tail
        bp = rd.pos;
        if (bp < 0) return ZL_E_CORRUPTION;
    } else {
        bp -= (i32)ll_log;
        sl = zl_bits_bwd(bs, bn, bp, ll_log);
        bp -= (i32)of_log;
        so = zl_bits_bwd(bs, bn, bp, of_log);
        bp -= (i32)ml_log;
        sm = zl_bits_bwd(bs, bn, bp, ml_log);
        if (bp < 0) return ZL_E_CORRUPTION;
    }
    return seq_tail(S, L, r, bs, bn, bp, i, sl, so, sm, r0, r1, r2, lit_used, out_used, oseq);
}


// ---- the same chain on a QUAD of lanes (launches of single-block frames, where a wave has lanes to spare):
// lane 0 of the quad owns the literal-length state, lane 1 the offset state, lane 2 the match-length
// state.  Per sequence every lane does the same table-driven work for its own state -- one table
// entry, its extra bits and its state bits cut out of a 128-bit window of the stream that all four
// lanes hold -- after the three (extra bits, state bits) pairs have gone round the quad with
// quad-permute DPP moves (register moves, not LDS traffic), so that every lane knows where its
// fields sit.  The three values go round the same way; the repcode rules then run identically in
// all four lanes and lane 3 writes the record.  About half the instructions per sequence of the
// one-lane chain, whose cost is its length.
template <int K>
__device__ __forceinline__ u32 quad_get(u32 v) {  // v of lane K of this lane's quad
    return (u32)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xF, 0xF, true);
}
// n bits (0..32) whose lowest sits at bit `at` of the 128-bit window x (dwords x.x .. x.w, ascending)
typedef u32 zl_v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32 window_bits(const zl_v4u32 x, u32 at, u32 n) {
    // (plain arithmetic: written with ?: on the dword index the compiler builds a tree of exec-masked branches)
    const u64 lo64 = ((u64)x.y << 32) | x.x, hi64 = ((u64)x.w << 32) | x.z;
    const u32 sh = at & 63;
    const u64 low = (lo64 >> sh) | ((hi64 << 1) << (63 - sh));  // bits from `at` on when at < 64
    const u64 high = hi64 >> sh;                                  // ... when at >= 64
    const u64 pick = 0ull - (u64)(at >> 6);                       // all ones when at >= 64
    const u32 v = (u32)((high & pick) | (low & ~pick));
    const u32 mask = (u32)((1ull << n) - 1);
    return v & mask;
}
__device__ __forceinline__ u32 seq_chain_quad(SeqShared& S, const SeqLane& L, u32 lane, const zl2_blk_rec& r, const u8* sp, u32 pos,
                                              u32 ll_log, u32 of_log, u32 ml_log, bool symbolic, uint4* oseq) {
    const u32 role = lane & 3;
    u32 r0 = symbolic ? ZL2_REP_SYM : 1u, r1 = symbolic ? (ZL2_REP_SYM | 1u) : 4u, r2 = symbolic ? (ZL2_REP_SYM | 2u) : 8u;
    const u8* bs = sp + pos;
    const u32 bn = r.seq_len - pos;
    i32 bp = zl_bwd_init(bs, bn);
    if (bp < 0) return ZL_E_CORRUPTION;
    const u32 nseq = r.nseq;
    u32 lit_used = 0, out_used = 0, i = 0;
    bp -= (i32)ll_log;
    u32 sl = zl_bits_bwd(bs, bn, bp, ll_log);
    bp -= (i32)of_log;
    u32 so = zl_bits_bwd(bs, bn, bp, of_log);
    bp -= (i32)ml_log;
    u32 sm = zl_bits_bwd(bs, bn, bp, ml_log);
    if (bp < 0) return ZL_E_CORRUPTION;
    u32 bad = 0;
    if (bp >= 160) {
        const u32 mis = (u32)((uintptr_t)bs & 3);
        const u32* w = (const u32*)(bs - mis);     // aligned dwords; the cursor c counts bits from w[0]: unread bits are [mis * 8, c)
        u32 c = mis * 8 + (u32)bp;
        u32 s = role == 0 ? sl : (role == 1 ? so : sm);
        const u32 ebase = role == 0 ? 0u : (role == 1 ? 1024u : 512u);
        const u32* vtab = &S.val_tab[role == 3 ? 2 : role][0];
        u32 Dw = (c - 1) >> 5;                      // dword that holds the top unread bit; the window is w[Dw - 3 .. Dw]
        zl_v4u32 x = *(const zl_v4u32*)(w + Dw - 3);
        zl_v4u32* ring = (zl_v4u32*)&S.ring[0][lane];
        // The table entry of the NEXT sequence is asked for as soon as its state is known -- in front of the values, the
        // repcode rules and the record of this one -- and carried round the loop: the look-up is the loop's dependent
        // chain.  (volatile: left to itself the compiler sinks the whole state computation behind the record store,
        // because nothing in this iteration uses it, and the chain then runs end to end.)
        u32 m_a = role != 1 ? ~0u : 0u, m_b = (role == 0 || role == 3) ? ~0u : 0u, m_c = role != 0 ? ~0u : 0u, m_d = (role == 1 || role == 3) ? ~0u : 0u;
        asm volatile("" : "+v"(m_a), "+v"(m_b), "+v"(m_c), "+v"(m_d));  // (kept in vector registers: told what they hold, the compiler turns the ANDs back into selects)
        typedef __attribute__((address_space(3))) const volatile u32 lds_cvu32;  // (through a generic pointer a volatile load is a flat load with a wait behind it)
        const u32 ent_at = (u32)(uintptr_t)&L.ent[ebase];  // LDS byte offset of this lane's table
        u32 ce = *(lds_cvu32*)(uintptr_t)(ent_at + 4 * s);
        while (i < nseq && Dw >= 4) {
            const u32 sym = (ce >> 16) & 0xFF, e = ce >> 24, b = (ce >> 10) & 0xF;
            const u32 pk = e | (b << 8);
            const u32 p0 = quad_get<0>(pk), p1 = quad_get<1>(pk), p2 = quad_get<2>(pk);
            const u32 e_ll = p0 & 0xFF, b_ll = p0 >> 8, e_of = p1 & 0xFF, b_of = p1 >> 8, e_ml = p2 & 0xFF, b_ml = p2 >> 8;
            const u32 keep = (i + 1 == nseq) ? 0u : 0xFFFFFFFFu;  // the last sequence reads no state bits
            // stream order below the cursor: offset bits, match-length bits, literal-length bits, then the state bits LL, ML, OF
            const u32 e_all = e_of + e_ml + e_ll;
            const u32 T = e_all + ((b_ll + b_ml + b_of) & keep);
            // (per-lane masks instead of selects on the role: the selects' lane masks live in scalar registers, of
            // which this kernel has none to spare -- each cost a v_readlane per sequence)
            const u32 above_e = (m_a & e_of) + (m_b & e_ml);            // role 1: 0; role 2: e_of; roles 0, 3: e_of + e_ml
            const u32 above_s = e_all + (m_c & b_ll) + (m_d & b_ml);    // role 0: 0; role 2: b_ll; roles 1, 3: b_ll + b_ml
            const u32 top = c - ((Dw - 3) << 5);  // the cursor inside the window: 97..128
            const u32 extra = window_bits(x, top - above_e - e, e);
            const u32 u = window_bits(x, top - above_s - (b & keep), b & keep);
            s = (ce & 0x3FF) + u;
            ce = *(lds_cvu32*)(uintptr_t)(ent_at + 4 * s);
            c -= T;
            const u32 Dn = (c - 1) >> 5;
            // (the same address in all four lanes, asked for before the values are needed.  Dn >= 1 here, so the load may
            // start up to 8 bytes in front of the stream: inside the block, whose headers and literals precede it)
            const zl_v4u32 xn = *(const zl_v4u32*)(w + (i32)Dn - 3);
            const u32 val = vtab[sym] + extra;
            const u32 ll = quad_get<0>(val), ofv = quad_get<1>(val), ml = quad_get<2>(val);
            // repcode rules (zstd.c:44291-44327), the same in every lane; masks instead of ?: (see window_bits)
            const u32 m_real = 0u - (u32)(ofv > 3);
            const u32 ri = ofv - 1 + (u32)(ll == 0);  // meaningful when not real: 0..3
            const u32 m1 = 0u - (u32)(ri == 1), m2 = 0u - (u32)(ri == 2), m3 = 0u - (u32)(ri == 3);
            const u32 r0m = r0 + 5 * (r0 >> 31) - 1;  // rep_minus1: a symbolic history entry counts the decrements in its upper bits
            const u32 cand = (r1 & m1) | (r2 & m2) | (r0m & m3) | (r0 & ~(m1 | m2 | m3));
            const u32 off = ((ofv - 3) & m_real) | (cand & ~m_real);
            const u32 m_sh2 = m_real | m2 | m3, m_sh1 = m_sh2 | m1;
            r2 = (r1 & m_sh2) | (r2 & ~m_sh2);
            r1 = (r0 & m_sh1) | (r1 & ~m_sh1);
            r0 = off;
            bad |= (u32)(off == 0);
            const zl_v4u32 rec = {lit_used, out_used, off, 0u};
            lit_used += ll;
            out_used += ll + ml;
            bad |= (out_used > ZL_BLOCK_MAX) ? 1u : 0u;
            if (role == 3) {
                ring[(i & 7) * 64] = rec;
                if ((i & 7) == 7) {
                    zl_v4u32* o = (zl_v4u32*)(oseq + (i - 7));
#pragma unroll
                    for (int q = 0; q < 8; q++) o[q] = ring[q * 64];
                }
            }
            x = xn;
            Dw = Dn;
            i++;
        }
        if (role == 3) for (u32 k = i & ~7u; k < i; k++) { const zl_v4u32 v = ring[(k & 7) * 64]; oseq[k] = make_uint4(v.x, v.y, v.z, v.w); }
        bp = (i32)c - (i32)(mis * 8);
        sl = quad_get<0>(s);
        so = quad_get<1>(s);
        sm = quad_get<2>(s);
    }
    if (bad) return ZL_E_CORRUPTION;
    if (bp < 0) return ZL_E_CORRUPTION;
    if (role != 3) return 0;  // the rest of the chain, the checks at its end and the totals: one lane
    return seq_tail(S, L, r, bs, bn, bp, i, sl, so, sm, r0, r1, r2, lit_used, out_used, oseq);
}

// ---- the quad chain on TWO waves (round 5; fuse1 bit 2).  A chain's time is its instruction count: a wave issues one vector
// instruction every four cycles whatever the number of lanes at work, and of the 150 instructions of a step above only about 85
// belong to what the NEXT step waits for -- the state's table entry, the bit cursor, the state bits.  The other 65 -- the three
// values round the quad, the repeat-offset rules, the running totals, the record and its ring -- depend on nothing but the values.
// So the workgroup has a second wave: wave 0 (PRODUCER) runs states and cursor and leaves every step's three values in LDS, sixteen
// steps to a tile; wave 1 (CONSUMER, one lane per chain) follows a tile behind, turns values into records and, at the end of a
// chain, runs the generic tail (seq_tail) from where the producer stopped.  Two tiles; one workgroup barrier per tile (the producer
// may not start tile k + 2 before the consumer has left tile k, which is what the barrier of tile k + 1 says).  The tiles, the
// consumer's record ring and the hand-over words live in S.ring, which the table builds have left by then.
#define ZLQ_TS 16u
struct QuadFin { u32 i; i32 bp; u32 sl, so, sm, err, act, pad; };
struct QuadShare {                 // laid over S.ring (8 KiB)
    uint4 ring[8][SEQ_FRAMES];     // the consumer's finished records: eight per chain, slot-major
    u32 tile[2][ZLQ_TS][32];       // [tile][step][producer lane]: the step's value of that lane's role (LL, offset, ML; role 3: unused)
    u32 tile_n[2][SEQ_FRAMES];     // steps of each chain in the tile
    u32 more[2];                   // the producer goes on behind this tile
    QuadFin fin[SEQ_FRAMES];       // where the producer stopped: sequence index, bit cursor, the three states; or its error
};
static_assert(sizeof(QuadShare) <= sizeof(uint4) * 8 * 64, "the hand-over fits the record ring's LDS");

__device__ __forceinline__ void quad_produce(const zl2_args& a, SeqShared& S, u32 wg, u32 lane) {
    QuadShare& Q = *(QuadShare*)&S.ring[0][0];
    const u32 fq = (lane >> 2) & (SEQ_FRAMES - 1), role = lane & 3;
    bool act = lane < 4 * SEQ_FRAMES && S.qpos[fq] != 0xFFFFFFFFu;
    u32 err = 0, nseq = 0, i = 0, sl = 0, so = 0, sm = 0;
    i32 bp = 0;
    const u8* bs = nullptr;
    u32 bn = 0;
    if (act) {
        const u32 f4 = a.order[wg * SEQ_FRAMES + fq];
        const zl2_blk_rec r4 = a.brec[(size_t)f4 * ZL2_MAXB];
        const u32 pos = S.qpos[fq];
        bs = a.src + a.frames[f4].src_off + r4.src_off + r4.seq_off + pos;
        bn = r4.seq_len - pos;
        nseq = r4.nseq;
        const u32 ll_log = S.logs[fq][0], of_log = S.logs[fq][1], ml_log = S.logs[fq][2];
        bp = zl_bwd_init(bs, bn);
        if (bp < 0) { err = ZL_E_CORRUPTION; act = false; }
        else {
            bp -= (i32)ll_log; sl = zl_bits_bwd(bs, bn, bp, ll_log);
            bp -= (i32)of_log; so = zl_bits_bwd(bs, bn, bp, of_log);
            bp -= (i32)ml_log; sm = zl_bits_bwd(bs, bn, bp, ml_log);
            if (bp < 0) { err = ZL_E_CORRUPTION; act = false; }
        }
    }
    const bool fast = act && bp >= 160;
    // (everything the loop carries; meaningful in the lanes of a fast chain)
    const u32 mis = fast ? (u32)((uintptr_t)bs & 3) : 0u;
    const u32* w = (const u32*)(bs - mis);
    u32 c = mis * 8 + (u32)bp;
    u32 s = role == 0 ? sl : (role == 1 ? so : sm);
    const u32 ebase = role == 0 ? 0u : (role == 1 ? 1024u : 512u);
    const u32* vtab = &S.val_tab[role == 3 ? 2 : role][0];
    u32 Dw = fast ? (c - 1) >> 5 : 0u;
    zl_v4u32 x = {0u, 0u, 0u, 0u};
    if (fast) x = *(const zl_v4u32*)(w + Dw - 3);
    u32 m_a = role != 1 ? ~0u : 0u, m_b = (role == 0 || role == 3) ? ~0u : 0u, m_c = role != 0 ? ~0u : 0u, m_d = (role == 1 || role == 3) ? ~0u : 0u;
    asm volatile("" : "+v"(m_a), "+v"(m_b), "+v"(m_c), "+v"(m_d));
    typedef __attribute__((address_space(3))) const volatile u32 lds_cvu32;
    const u32 ent_at = (u32)(uintptr_t)&S.L[fq].ent[ebase];
    u32 ce = fast ? *(lds_cvu32*)(uintptr_t)(ent_at + 4 * s) : 0u;
    bool told = false;
    for (u32 k = 0;; k++) {
        u32 n = 0;
        if (fast) {
            while (n < ZLQ_TS && i < nseq && Dw >= 4) {
                const u32 sym = (ce >> 16) & 0xFF, e = ce >> 24, b = (ce >> 10) & 0xF;
                const u32 pk = e | (b << 8);
                const u32 p0 = quad_get<0>(pk), p1 = quad_get<1>(pk), p2 = quad_get<2>(pk);
                const u32 e_ll = p0 & 0xFF, b_ll = p0 >> 8, e_of = p1 & 0xFF, b_of = p1 >> 8, e_ml = p2 & 0xFF, b_ml = p2 >> 8;
                const u32 keep = (i + 1 == nseq) ? 0u : 0xFFFFFFFFu;  // the last sequence reads no state bits
                const u32 e_all = e_of + e_ml + e_ll;
                const u32 T = e_all + ((b_ll + b_ml + b_of) & keep);
                const u32 above_e = (m_a & e_of) + (m_b & e_ml);
                const u32 above_s = e_all + (m_c & b_ll) + (m_d & b_ml);
                const u32 top = c - ((Dw - 3) << 5);
                const u32 u = window_bits(x, top - above_s - (b & keep), b & keep);
                s = (ce & 0x3FF) + u;
                ce = *(lds_cvu32*)(uintptr_t)(ent_at + 4 * s);
                const u32 extra = window_bits(x, top - above_e - e, e);
                c -= T;
                const u32 Dn = (c - 1) >> 5;
                const zl_v4u32 xn = *(const zl_v4u32*)(w + (i32)Dn - 3);
                Q.tile[k & 1][n][lane] = vtab[sym] + extra;
                x = xn;
                Dw = Dn;
                i++;
                n++;
            }
        }
        const bool more = fast && i < nseq && Dw >= 4;
        if (fast && !more && !told) {  // this chain's fast part is over: where the tail takes up
            bp = (i32)c - (i32)(mis * 8);
            sl = quad_get<0>(s);
            so = quad_get<1>(s);
            sm = quad_get<2>(s);
        }
        if (lane < 4 * SEQ_FRAMES && role == 0) {
            Q.tile_n[k & 1][fq] = n;
            if (!more && !told) { QuadFin fn; fn.i = i; fn.bp = bp; fn.sl = sl; fn.so = so; fn.sm = sm; fn.err = err; fn.act = (act || err) ? 1u : 0u; fn.pad = 0; Q.fin[fq] = fn; }
        }
        told = told || !more;
        const bool any_more = __any(more);
        if (lane == 0) Q.more[k & 1] = any_more ? 1u : 0u;
        __syncthreads();  // tile k is the consumer's
        if (!any_more) break;
    }
}

__device__ __forceinline__ void quad_consume(const zl2_args& a, SeqShared& S, u32 wg, u32 lane /* of wave 1 */) {
    QuadShare& Q = *(QuadShare*)&S.ring[0][0];
    const u32 ch = lane & (SEQ_FRAMES - 1);
    const bool mine = lane < SEQ_FRAMES && S.qpos[ch] != 0xFFFFFFFFu;
    u32 f4 = 0;
    zl2_blk_rec r4;
    r4.type = 0; r4.nseq = 0; r4.seq_len = 0; r4.lit_size = 0;
    uint4* oseq = nullptr;
    bool symbolic = false;
    if (mine) {
        f4 = a.order[wg * SEQ_FRAMES + ch];
        r4 = a.brec[(size_t)f4 * ZL2_MAXB];
        oseq = (uint4*)a.seq_scratch + a.plan[f4].seq_off + r4.seq_dst;
        symbolic = starts_symbolic(a, f4);
    }
    u32 r0 = symbolic ? ZL2_REP_SYM : 1u, r1 = symbolic ? (ZL2_REP_SYM | 1u) : 4u, r2 = symbolic ? (ZL2_REP_SYM | 2u) : 8u;
    u32 lit_used = 0, out_used = 0, i = 0, bad = 0;
    zl_v4u32* ring = (zl_v4u32*)&Q.ring[0][ch];
    for (u32 k = 0;; k++) {
        __syncthreads();  // tile k is ready
        if (mine) {
            const u32 n = Q.tile_n[k & 1][ch];
            for (u32 j = 0; j < n; j++) {
                const u32 ll = Q.tile[k & 1][j][4 * ch], ofv = Q.tile[k & 1][j][4 * ch + 1], ml = Q.tile[k & 1][j][4 * ch + 2];
                // repcode rules (zstd.c:44291-44327); masks instead of ?: (see window_bits)
                const u32 m_real = 0u - (u32)(ofv > 3);
                const u32 ri = ofv - 1 + (u32)(ll == 0);
                const u32 m1 = 0u - (u32)(ri == 1), m2 = 0u - (u32)(ri == 2), m3 = 0u - (u32)(ri == 3);
                const u32 r0m = r0 + 5 * (r0 >> 31) - 1;
                const u32 cand = (r1 & m1) | (r2 & m2) | (r0m & m3) | (r0 & ~(m1 | m2 | m3));
                const u32 off = ((ofv - 3) & m_real) | (cand & ~m_real);
                const u32 m_sh2 = m_real | m2 | m3, m_sh1 = m_sh2 | m1;
                r2 = (r1 & m_sh2) | (r2 & ~m_sh2);
                r1 = (r0 & m_sh1) | (r1 & ~m_sh1);
                r0 = off;
                bad |= (u32)(off == 0);
                const zl_v4u32 rec = {lit_used, out_used, off, 0u};
                lit_used += ll;
                out_used += ll + ml;
                bad |= (out_used > ZL_BLOCK_MAX) ? 1u : 0u;
                ring[(i & 7) * SEQ_FRAMES] = rec;
                if ((i & 7) == 7) {
                    zl_v4u32* o = (zl_v4u32*)(oseq + (i - 7));
#pragma unroll
                    for (int q = 0; q < 8; q++) o[q] = ring[q * SEQ_FRAMES];
                }
                i++;
            }
        }
        if (!Q.more[k & 1]) break;
    }
    if (mine) {
        for (u32 k = i & ~7u; k < i; k++) { const zl_v4u32 v = ring[(k & 7) * SEQ_FRAMES]; oseq[k] = make_uint4(v.x, v.y, v.z, v.w); }
        const QuadFin fn = Q.fin[ch];
        u32 err = fn.err;
        if (!err && fn.act) {
            if (bad || fn.bp < 0 || fn.i != i) err = ZL_E_CORRUPTION;
            else {
                const u32 pos = S.qpos[ch];
                const u8* bs = a.src + a.frames[f4].src_off + r4.src_off + r4.seq_off + pos;
                err = seq_tail(S, S.L[ch], r4, bs, r4.seq_len - pos, fn.bp, i, fn.sl, fn.so, fn.sm, r0, r1, r2, lit_used, out_used, oseq);
            }
        }
        if (err) set_err(a.frec + f4, err);
    }
}

__device__ __forceinline__ void wave_sync() {  // the sequence role runs on one wave: LDS order is program order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One lane per (frame, block): lane = block * SEQ_FRAMES + frame.  A block that describes at least
// one table is the "leader" of the blocks behind it that repeat all three (mode byte 0xFC);
// round r builds block r's tables, then every lane whose leader is r runs its chain.  Frames from
// the HIP encoder share one table set per frame and finish in one round with all lanes busy.
// QUAD (launches of single-block frames): the chain of a frame runs on four lanes, seq_chain_quad.
// Barriers: this role is ONE wave and orders its LDS traffic with wave_sync().  The only workgroup barriers (__syncthreads) it may hold are
// the two-wave chains' protocol -- cmd = 1 / barrier / quad_produce / barrier per chain phase, cmd = 0 / barrier at the end -- because
// zl2_entropy1_kernel's second wave counts them; no path may leave the role in front of the last one.
template <bool QUAD>
__device__ __forceinline__ void seq_role(const zl2_args& a, SeqShared& S, u32* rounds, u32 wg) {
    const u32 lane = threadIdx.x;
    if (lane >= 64) return;
    if (QUAD) {
        if (lane < 36) S.val_tab[0][lane] = k_LL_base[lane];
        if (lane < 32) S.val_tab[1][lane] = 1u << lane;
        if (lane < 53) S.val_tab[2][lane] = k_ML_base[lane];
    }
    if (lane < 36) S.ll_tab[lane] = k_LL_base[lane] | ((u32)k_LL_bits[lane] << 24);
    if (lane < 53) S.ml_tab[lane] = k_ML_base[lane] | ((u32)k_ML_bits[lane] << 24);
    if (lane < 36) S.defn[0][lane] = k_LL_default[lane];
    if (lane < 29) S.defn[1][lane] = k_OF_default[lane];
    if (lane < 53) S.defn[2][lane] = k_ML_default[lane];
    if (lane < SEQ_FRAMES) { S.logs[lane][0] = 0; S.logs[lane][1] = 0; S.logs[lane][2] = 0; S.logs[lane][3] = 0; }
    if (lane == 0) *rounds = 0;
    wave_sync();
    const u32 fi = lane % SEQ_FRAMES, b = lane / SEQ_FRAMES;
    const u32 slot = wg * SEQ_FRAMES + fi;
    const u32 f = slot < a.nframes ? a.order[slot] : 0xFFFFFFFFu;
    bool active = false;
    int lb = -1;
    zl2_blk_rec r;
    r.type = 0;
    const u8* src = nullptr;
    uint4* q_seq = nullptr;
    zl2_frame_rec* fr = nullptr;
    if (f < a.nframes) {
        fr = a.frec + f;
        const u32 nblocks = (fr->err == 0 && !fr->skippable) ? fr->nblocks : 0;
        if (b < nblocks) {
            const zl2_blk_rec* br = a.brec + (size_t)f * ZL2_MAXB;
            src = a.src + a.frames[f].src_off;
            q_seq = (uint4*)a.seq_scratch + a.plan[f].seq_off;  // {literals before, output bytes before, offset, -}
            for (u32 k = 0; k <= b; k++)
                if (br[k].type == 2 && br[k].nseq > 0 && br[k].seq_modes != 0xFCu) lb = (int)k;
            r = br[b];
            active = (r.type == 2 && r.nseq > 0);
            if (active && r.seq_modes == 0x100u) { set_err(fr, ZL_E_SRCSIZE_WRONG); active = false; }
            else if (active && lb < 0 && !(a.ext != nullptr && ext_rec(a, f, 1).type == 2)) { set_err(fr, ZL_E_CORRUPTION); active = false; }  // repeat mode with nothing to repeat (zstd.c:43666)
        }
    }
    // Rounds: 0..2 build the tables a segment of a large frame inherits from earlier blocks (large-frame
    // pipeline only; no chains), 3 runs the chains of blocks that repeat exactly those, 4 + k is led by
    // block k of the frame (segment).
    u32 ext_n = 0;  // inherited table descriptions of this lane's segment (block 0's lane reads them)
    if (a.ext != nullptr && f < a.nframes && b == 0 && fr->err == 0 && !fr->skippable && fr->nblocks != 0)
        for (u32 k = 1; k < 4; k++) if (ext_rec(a, f, k).type == 2) ext_n = k;
    if (f < a.nframes && src == nullptr) src = a.src + a.frames[f].src_off;
    if (active) atomicOr(rounds, 1u << (lb + 4));
    if (ext_n) atomicOr(rounds, (1u << ext_n) - 1);
    wave_sync();
    const u32 mask = *rounds;
    SeqLane& L = S.L[fi];
    const bool prof = ZL_ROLE_STAMPS && a.stats != nullptr && lane == 0;
    u64 p_build = 0, p_chain = 0, p_parse = 0, p_tab = 0;
    for (u32 ri = 0; ri < 4 + ZL2_MAXB; ri++) {
        if (!((mask >> ri) & 1)) continue;
        const u32 round = ri;  // (stamps the parsed header)
        const u64 q0 = prof ? __builtin_amdgcn_s_memtime() : 0;
        const bool mine = active && lb + 4 == (int)ri;
        const bool ext_lead = ri < 3 && ri < ext_n;  // block 0's lane stands in for the earlier block
        u32 pos = 1;  // a block that repeats all tables: the chain starts right after the mode byte
        SeqBuild& B = *(SeqBuild*)((u8*)S.ring + fi * 1024);
        // the scratch shares LDS with the record ring: whatever the last chains left there must not look like a parsed header
        if (b == 0) B.stamp = 0;
        wave_sync();
        if (ext_lead && a.dict_pre != nullptr) {
            // the dictionary's three tables, built once per context (zl2_dict_prebuild_kernel): copied, nothing to parse or build
            const zl2_dict_pre* P = a.dict_pre;
            if (P->seq_ok == 1) {
                const uint4* from = (const uint4*)P->ent;
                uint4* to = (uint4*)L.ent;
                for (u32 i = 0; i < 1280 * 4 / 16; i += 8) {
                    uint4 v[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = from[i + q];
#pragma unroll
                    for (int q = 0; q < 8; q++) to[i + q] = v[q];
                }
                S.logs[fi][0] = P->seq_log[0]; S.logs[fi][1] = P->seq_log[1]; S.logs[fi][2] = P->seq_log[2];
                if (S.logs[fi][3] != 2) S.logs[fi][3] = 1;
            } else { set_err(fr, ZL_E_DICT_CORRUPTED); S.logs[fi][3] = 2; }
        } else
        if (ext_lead || (mine && b + 4 == ri)) {  // leader: copy the table descriptions out of the frame, then parse them
            zl2_blk_rec lr = r;
            if (ext_lead) lr = ext_rec(a, f, 1 + ri);
            const u8* sp = (ext_lead ? ext_src(a, f, lr) : src) + lr.src_off + lr.seq_off;
            const u32 sn = lr.seq_len, hn = sn < sizeof(B.hdr) ? sn : (u32)sizeof(B.hdr);
            {   // whole 16-byte pieces in one round trip, the last few bytes one by one
                const u32 full = hn >> 4;
                uint4 v[sizeof(B.hdr) / 16];
#pragma unroll
                for (u32 q = 0; q < sizeof(B.hdr) / 16; q++) if (q < full) __builtin_memcpy(&v[q], sp + q * 16, 16);
#pragma unroll
                for (u32 q = 0; q < sizeof(B.hdr) / 16; q++) if (q < full) *(uint4*)(B.hdr + q * 16) = v[q];
                for (u32 x = full * 16; x < hn; x++) B.hdr[x] = sp[x];
            }
            const u32 modes = B.hdr[0];
            const bool have_tables = S.logs[fi][3] == 1 || ext_lead;  // (an inherited description may itself repeat a table: the block that defines that one is another of the three)
            bool ok = (modes & 3) == 0;
            if (ok) {
                u32 used = parse_seq_table(B, 0, modes >> 6, B.hdr + pos, hn - pos, have_tables);
                ok = used != 0xFFFFFFFFu;
                if (ok) { pos += used; used = parse_seq_table(B, 1, (modes >> 4) & 3, B.hdr + pos, hn - pos, have_tables); ok = used != 0xFFFFFFFFu; }
                if (ok) { pos += used; used = parse_seq_table(B, 2, (modes >> 2) & 3, B.hdr + pos, hn - pos, have_tables); ok = used != 0xFFFFFFFFu; }
                if (ok) pos += used;
            }
            // descriptions that may reach past the copy go to the general kernel, which reads them in place
            if (hn < sn && (!ok || pos + 8 > hn)) { set_err(fr, ZL_E_FALLBACK); atomicAdd(a.max_blocks + 1, 1u); ok = false; S.logs[fi][3] = 2; }
            else if (!ok) { set_err(fr, ZL_E_CORRUPTION); S.logs[fi][3] = 2; }
            B.stamp = ok ? round + 1 : 0;
        }
        wave_sync();
        const u64 qp = prof ? __builtin_amdgcn_s_memtime() : 0;
        if (b < 3 && f < a.nframes && B.stamp == round + 1) {  // lanes of blocks 0..2 build one table each (kind 0 LL, 1 OF, 2 ML)
            const u32 lg = make_seq_table(L, B, &S.defn[0][0], (int)b, S.logs[fi][b]);
            if (lg == 0xFFFFFFFFu) { set_err(fr, ZL_E_CORRUPTION); S.logs[fi][3] = 2; }
            else S.logs[fi][b] = lg;
        }
        const u64 qb = prof ? __builtin_amdgcn_s_memtime() : 0;
        wave_sync();
        if (b == 0 && f < a.nframes && B.stamp == round + 1) { if (S.logs[fi][3] != 2) S.logs[fi][3] = 1; B.stamp = 0; }
        wave_sync();
        const u64 q1 = prof ? __builtin_amdgcn_s_memtime() : 0;
        if (QUAD) {
            // the frame's own lane says where its bitstream starts; the quad lanes 4 * slot .. 4 * slot + 3 run the chain
            if (b == 0) S.qpos[fi] = (mine && S.logs[fi][3] == 1 && fr->err == 0) ? pos : 0xFFFFFFFFu;
            wave_sync();
            const u32 fq = lane >> 2;
            if (a.fuse1 & 4) {  // the chains on two waves: this one produces, wave 1 consumes (quad_produce / quad_consume)
                if (lane == 0) S.cmd = 1;
                __syncthreads();  // wave 1 wakes up
                quad_produce(a, S, wg, lane);
                __syncthreads();  // ... and has written its records, totals and errors
            } else
            if (lane < 4 * SEQ_FRAMES && S.qpos[fq] != 0xFFFFFFFFu) {
                const u32 f4 = a.order[wg * SEQ_FRAMES + fq];
                const zl2_blk_rec r4 = a.brec[(size_t)f4 * ZL2_MAXB];
                const u8* sp4 = a.src + a.frames[f4].src_off + r4.src_off + r4.seq_off;
                uint4* o4 = (uint4*)a.seq_scratch + a.plan[f4].seq_off + r4.seq_dst;
                const u32 err = seq_chain_quad(S, S.L[fq], lane, r4, sp4, S.qpos[fq], S.logs[fq][0], S.logs[fq][1], S.logs[fq][2], starts_symbolic(a, f4), o4);
                if (err) set_err(a.frec + f4, err);
            }
        } else if (mine && S.logs[fi][3] == 1 && fr->err == 0) {
            const u32 err = seq_chain(S, L, lane, r, src + r.src_off + r.seq_off, pos, S.logs[fi][0], S.logs[fi][1], S.logs[fi][2], b != 0 || starts_symbolic(a, f),
                                      q_seq + r.seq_dst);
            if (err) set_err(fr, err);
        }
        wave_sync();
        if (prof) { const u64 q2 = __builtin_amdgcn_s_memtime(); p_build += q1 - q0; p_chain += q2 - q1; p_parse += qp - q0; p_tab += qb - qp; }
    }
    if (QUAD && (a.fuse1 & 4)) {  // no more chains: wave 1 may go
        if (lane == 0) S.cmd = 0;
        __syncthreads();
    }
    // profile (slots of the general kernel's phase array, unused when no frame falls back): table builds, chains
    if (prof) {
        atomicAdd((unsigned long long*)&a.stats[-8], (unsigned long long)p_build);
        atomicAdd((unsigned long long*)&a.stats[-7], (unsigned long long)p_chain);
        atomicAdd((unsigned long long*)&a.stats[-6], 1ull);
        atomicAdd((unsigned long long*)&a.stats[-5], (unsigned long long)p_parse);
        atomicAdd((unsigned long long*)&a.stats[-4], (unsigned long long)p_tab);
    }
}

// Literal lanes and sequence lanes are independent, want different shapes (256 threads and 69 KiB
// of LDS against one wave and 50 KiB) and each under-fills the chip on its own, so they are two
// kernels launched side by side on two streams.  Measured on 10 000 reference-made 64 KiB frames,
// whose single long sequence chain per frame is the critical resource: fused into one launch (one
// workgroup shape for both roles) the two roles took 22.6 ms, side by side 8-9 ms.
static_assert(LIT_FRAMES == SEQ_FRAMES, "both roles walk the frames in the same groups");
#ifndef ZL_LIT_PAD_BYTES
#define ZL_LIT_PAD_BYTES (22 << 10)
#endif
template <int MB>
__global__ __launch_bounds__(LIT_FRAMES * MB * 4, 1) void zl2_lit_kernel(zl2_args a) {
    __shared__ LitGroupT<MB> G[LIT_FRAMES];
    __shared__ u32 rounds;  // bit r: some lane of this workgroup has block r as its leader
    // both instantiations are launched; the index kernel's verdict says which one works
    if ((*a.max_blocks > 1) != (MB > 1)) return;
    if (MB == 1 && a.fuse1) return;  // (zl2_entropy1_kernel does the work)
    // Resource shaping (general instantiation): with 22 KiB of dynamic LDS it never touches a CU
    // takes ONE literal workgroup plus one sequence workgroup (or three sequence workgroups where no
    // literal workgroup is left) instead of two literal workgroups that would keep the sequence
    // chains -- the longer pole -- waiting.
    ZL_POISON(G, 1);
    const bool prof = ZL_ROLE_STAMPS && a.stats != nullptr && threadIdx.x == 0;
    const long long t0 = prof ? clock64() : 0;
    u32 g0, g1; bool split;
    zl2_part_range(a, g0, g1, split);
    if (g0 + blockIdx.x >= g1) return;
    lit_role<MB>(a, G, &rounds, g0 + blockIdx.x);
    if (prof) atomicAdd((unsigned long long*)&a.stats[6], (unsigned long long)(clock64() - t0));  // profile: cycles per literal workgroup
}
// A dictionary's tables in the lanes' formats, once per context (zl_kernels.h zl2_dict_pre): lane 0 builds the Huffman table, then
// parses the synthetic block's sequences header; lanes 0..2 build one FSE table each -- the same functions the inheritance
// rounds run, so that a copy of the result is what they would have built.
__global__ __launch_bounds__(64) void zl2_dict_prebuild_kernel(const zl2_blk_rec* ext, const u8* ext_base, zl2_dict_pre* out) {
    __shared__ LitGroupT<1> G;
    __shared__ SeqLane L;
    __shared__ SeqBuild B;
    __shared__ short defn[3][64];
    __shared__ u32 logs[4];
    const u32 lane = threadIdx.x;
    if (lane < 36) defn[0][lane] = k_LL_default[lane];
    if (lane < 29) defn[1][lane] = k_OF_default[lane];
    if (lane < 53) defn[2][lane] = k_ML_default[lane];
    if (lane < 4) logs[lane] = 0;
    __syncthreads();
    if (lane == 0) {
        const zl2_blk_rec h = ext[0];
        const u32 tree = h.type == 2 ? build_huf_table(G, ext_base + h.src_off + h.lit_a, h.lit_b) : 0u;
        out->huf_ok = tree == 0xFFFFFFFFu ? 2u : (tree != 0 ? 1u : 0u);
        out->huf_log = G.log;
        const zl2_blk_rec q = ext[1];
        bool ok = q.type == 2 && q.seq_len >= 1 && q.seq_len <= sizeof(B.hdr);
        if (ok) {
            const u8* sp = ext_base + q.src_off + q.seq_off;
            for (u32 x = 0; x < q.seq_len; x++) B.hdr[x] = sp[x];
            const u32 modes = B.hdr[0];
            ok = modes == 0xA8u;
            u32 pos = 1;
            for (int k = 0; ok && k < 3; k++) {  // block order: LL, OF, ML (kinds 0, 1, 2)
                const u32 used = parse_seq_table(B, k, 2, B.hdr + pos, q.seq_len - pos, false);
                ok = used != 0xFFFFFFFFu;
                pos += ok ? used : 0;
            }
            ok = ok && pos == q.seq_len;
        }
        logs[3] = ok ? 1u : 0u;
    }
    __syncthreads();
    if (lane < 3 && logs[3]) {
        const u32 lg = make_seq_table(L, B, &defn[0][0], (int)lane, 0u);
        if (lg == 0xFFFFFFFFu) atomicExch(&logs[3], 0u);
        else logs[lane] = lg;
    }
    __syncthreads();
    if (lane == 0) { out->seq_ok = logs[3]; out->seq_log[0] = logs[0]; out->seq_log[1] = logs[1]; out->seq_log[2] = logs[2]; }
    for (u32 i = lane; i < 2048; i += 64) out->tab[i] = G.tab[i];
    for (u32 i = lane; i < 1280; i += 64) out->ent[i] = L.ent[i];
}

__global__ __launch_bounds__(64) void zl2_seq_kernel(zl2_args a) {
    __shared__ SeqShared S;
    __shared__ u32 rounds;
    if (a.fuse1 && *a.max_blocks <= 1) return;  // (zl2_entropy1_kernel does the work)
    ZL_POISON(S, 1);
    const bool prof = ZL_ROLE_STAMPS && a.stats != nullptr && threadIdx.x == 0;
    const long long t0 = prof ? clock64() : 0;
    u32 g0, g1; bool split;
    zl2_part_range(a, g0, g1, split);
    if (g0 + blockIdx.x >= g1) return;
    seq_role<false>(a, S, &rounds, g0 + blockIdx.x);
    if (prof) atomicAdd((unsigned long long*)&a.stats[7], (unsigned long long)(clock64() - t0));  // profile: cycles per sequence workgroup
}

// Launches whose frames all have ONE block (config C2's 64 KiB frames, the segments of a large frame):
// both roles then want nearly the same shape -- one wave with 50 KiB of LDS for eight sequence chains,
// half a wave with 41 KiB for eight frames' Huffman streams -- and the only thing two kernels on two
// streams add is a fight for the CUs' LDS (one 1 GiB reference frame: sequence kernel alone 15.2 ms,
// literal kernel alone 11.0 ms, side by side 19.9 ms, because a CU that has taken two literal
// workgroups has room for one sequence workgroup only, and the sequence chains are the longer pole).
// Fused, the workgroups are dealt out in the pattern sequence, sequence, literal -- each list sorted
// by decreasing cost -- which is what fits a CU: 2 x 50 + 41 KiB.
__global__ __launch_bounds__(128) void zl2_entropy1_kernel(zl2_args a) {
    __shared__ union U_ {
        SeqShared s;
        LitGroupT<1> g[LIT_FRAMES];
        __device__ U_() {}
    } U;
    __shared__ u32 rounds;
    if (!a.fuse1 || *a.max_blocks > 1) return;
    u32 g0, g1; bool split;
    zl2_part_range(a, g0, g1, split);  // the part of the cost-sorted order this launch covers
    const u32 r = blockIdx.x % 3, g = blockIdx.x / 3;
    // The consumer below counts workgroup barriers in step with seq_role's and reads `cmd` behind each: the word starts out as "the role
    // is over", so that a producer that ever ended without its last barrier (a wave that has ended no longer counts for a barrier) would
    // release the consumer into a break, not into a chain phase on whatever the LDS held
    if (r < 2) { if (threadIdx.x == 0) U.s.cmd = 0; __syncthreads(); }
    if (threadIdx.x >= 64) {
        // wave 1: the consumer of a sequence workgroup's two-wave chains; everywhere else it has nothing to do (a wave that has
        // ended no longer counts for the workgroup's barriers)
        const u32 sg = g0 + 2 * g + r;
        if (r < 2 && sg < g1 && (a.fuse1 & 6) == 6) {
            for (;;) {
                __syncthreads();
                if (U.s.cmd == 0) break;
                quad_consume(a, U.s, sg, threadIdx.x - 64);
                __syncthreads();
            }
        }
        return;
    }
    if (r < 2) {
        const u32 sg = g0 + 2 * g + r;
        if (sg < g1) { if (a.fuse1 & 2) seq_role<true>(a, U.s, &rounds, sg); else seq_role<false>(a, U.s, &rounds, sg); }
    } else if (g0 + g < g1) {
        lit_role<1>(a, U.g, &rounds, g0 + g);
    }
}

// ---------------------------------------------------------------------------------------
// zl2_exec_kernel: one workgroup per frame; only the batch being built lives in LDS.
//
// The sequence lanes left running positions (literals / output bytes before each sequence), so a
// batch needs no scan.  A batch is as many sequences as fit EX_CAP output bytes.  Per batch:
//   pass 1  the batch's literal bytes are scattered into the LDS image with coalesced dword loads;
//   pass 2a every match byte takes ONE hop through the batch table (owner-per-32-byte-granule
//           lookup): a hop that lands in front of the batch reads the byte from the output already
//           in memory, a hop onto a literal of the batch copies it, anything else is stored as a
//           16-bit link;
//   pass 2b links are shortened by pointer jumping (link := link of link) until every byte has
//           found an existing byte -- log2(chain length) rounds of plain LDS traffic instead of a
//           per-byte walk whose tail is as long as span/offset;
//   flush   the image goes to memory with 16-byte stores (it is laid out congruent to the
//           destination modulo 16) and becomes the "output already in memory" of the next batch.
// A sequence longer than EX_CAP is copied memory to memory: all of its sources lie in front of its
// match (zstd.c:44013 ZSTD_execSequence's overlap rule).  With 37 KiB of LDS per workgroup four
// frames share a CU (the earlier whole-frame image allowed one), which hides the barrier and
// memory latencies of one frame behind the work of the others.
// ---------------------------------------------------------------------------------------
#define EX_THREADS 512
#define EX_GRAN_LOG 5
#define EX_CAP 8192u     // output bytes per batch (links are 16-bit, relative to the batch start)
#define EX_MAXSEQ 1023u  // sequences per batch (+1 look-ahead entry)
// A byte's link inside a batch: an index below EX_CAP = the byte of the batch it copies (pending); EX_FINAL = its value stands in
// the image.  Large-frame pipeline (BIG) only: any other value v = the byte WAITS for frame position v - EX_CAP, which lies in
// another run of segments (or is itself waiting).  Small frames keep 16-bit links (38 KiB of LDS, four frames per CU); the large-frame
// instantiation has 32-bit ones -- round 2 kept 16-bit links PLUS a 32-bit position per byte there (71 KiB: two workgroups per CU;
// now 53 KiB: three).
template <bool BIG> struct ExLink { typedef u16 type; };
template <> struct ExLink<true> { typedef u32 type; };
#define EX_MAXSEQ_BIG 895u  // (sequences per batch of the large-frame instantiation: three workgroups' LDS must fit a CU)
template <bool BIG>
struct ExecSharedT {
    typedef typename ExLink<BIG>::type link_t;
    static constexpr u32 MAXSEQ = BIG ? EX_MAXSEQ_BIG : EX_MAXSEQ;
    u8 img[EX_CAP + 16];  // byte d of the block sits at img[d + ibase]; (img + d + ibase) and (destination + d) agree modulo 16
    link_t link[EX_CAP];
    u32 b_out[MAXSEQ + 1];
    u32 b_lit[MAXSEQ + 1];
    u32 b_off[MAXSEQ + 1];
    u16 own[(EX_CAP >> EX_GRAN_LOG) + 8];
    u64 xxh_acc[4];
    u32 bcast[8];  // [0] xxh result, [1] batch size, [4..6] rotating "still pending" flags
    u32 brec[ZL2_MAXB * 16];  // the frame's block records, fetched once
    int err;
    u32 pad3;
};

// sequence of the current batch that owns block-relative output position p
template <class Shared>
__device__ __forceinline__ u32 owner_of(const Shared& S, u32 p, u32 gbase) {
    u32 k = S.own[(p - gbase) >> EX_GRAN_LOG];
    while (p >= S.b_out[k + 1]) k++;
    return k;
}

// workgroup-wide copy between global buffers of any alignment
__device__ __forceinline__ void ex_copy(u8* dst, const u8* src, u32 n) {
    const u32 t = threadIdx.x, body = n & ~15u;
    for (u32 i = t * 16; i < body; i += EX_THREADS * 16) {
        uint4 v;
        __builtin_memcpy(&v, src + i, 16);
        __builtin_memcpy(dst + i, &v, 16);
    }
    if (t < n - body) dst[body + t] = src[body + t];
}

// Large-frame pipeline (BIG): the workgroup executes one SEGMENT (1 to ZL2_MAXB blocks) of a frame whose
// other segments run at the same time.  A match that reaches in front of the segment cannot be copied
// yet: the byte is left "deferred" with the frame position it copies in a.big->link, everything that
// depends on it inherits that position, and zl3_resolve_kernel fills the bytes in afterwards.
__device__ __forceinline__ u32 ex_link_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One frame (BIG: one segment; `seg_start` is the first output position of the workgroup's run of
// segments: what lies in front of it belongs to other workgroups).  Returns false when the run must stop.
// (PROF: the in-kernel phase stamps of ZL_d_profile = 2, a template parameter: kept as a run-time test their
// accumulators cost registers in a kernel that is held to 64 of them -- 2.07 -> 1.92 ms without)
template <bool BIG, bool PROF>
__device__ __forceinline__ bool exec_frame(const zl2_args& a, ExecSharedT<BIG>& S, const u32 f, u64& seg_start, const bool first) {
    const u32 t = threadIdx.x;
    const bool wg_prof = PROF && a.stats != nullptr && t == 0;
    const u64 wg_t0 = wg_prof ? __builtin_amdgcn_s_memtime() : 0;
    zl2_frame_rec* fr = a.frec + f;
    const u32 ferr = fr->err;
    if (!BIG && (ferr == ZL_E_FALLBACK || ferr == ZL_E_DONE)) return false;  // the general kernel (or the large-frame pipeline) owns this frame
    if (BIG && (ferr || a.big->info->err)) return false;  // (zl3_offsets_kernel has collected the segments' errors)
    if (ferr || fr->skippable) {
        if (t == 0) a.results[f] = ferr ? ZL_ERR(ferr) : 0;
        return false;
    }
    u32* const glink = BIG ? a.big->link : nullptr;
    const u32 dict_len = BIG ? a.big->dict_len : a.dict_len;  // uniform
    const u8* const dict_end = BIG ? a.big->dict_end : a.dict_end;
    typedef typename ExecSharedT<BIG>::link_t link_t;
    constexpr link_t EX_FINAL = (link_t)~(link_t)0;
    constexpr u32 EX_MAXSEQ_T = ExecSharedT<BIG>::MAXSEQ;
    const zl_frame_desc fd = a.frames[f];
    const u8* src = a.src + fd.src_off;
    u8* dst = a.dst + fd.dst_off;
    const u64 cap = fd.dst_cap;
    const zl2_blk_rec* br = a.brec + (size_t)f * ZL2_MAXB;
    const u8* lit_base = a.lit_scratch + a.plan[f].lit_off;
    const uint4* q_seq = (const uint4*)a.seq_scratch + a.plan[f].seq_off;
    if (t == 0) S.err = 0;
    const u32 nblocks = fr->nblocks;
    static_assert(sizeof(zl2_blk_rec) == 64, "block records are staged as 16 dwords each");
    if (t < nblocks * 16) S.brec[t] = ((const u32*)br)[t];
    __syncthreads();
    const u32 seg_blk0 = BIG ? a.big->seg_first[f] : 0u;  // the segment's first block inside the frame
    u64 op = BIG ? (u64)a.big->blk_dst[seg_blk0] : 0;
    if (first) seg_start = op;
    u32 rep0 = a.dict_rep[0], rep1 = a.dict_rep[1], rep2 = a.dict_rep[2];  // repcode history at the start of the next block with sequences (zstd.c:41433; 1, 4, 8 unless a formatted dictionary says otherwise, zstd.c:42140)
    const u64 wg_t1 = wg_prof ? __builtin_amdgcn_s_memtime() : 0;
    // Output goes to memory with plain stores and is read back (by other waves) only in pass 2a of a
    // later batch, so the one place that drains the stores is the barrier in front of pass 2a:
    // literal-only and raw blocks never wait for their stores.
    for (u32 b = 0; b < nblocks; b++) {
        const zl2_blk_rec r = *(const zl2_blk_rec*)&S.brec[b * 16];
        const u8* blk = src + r.src_off;
        u8* bdst = dst + op;
        if (BIG) {  // the history this block starts with was worked out by zl3_offsets_kernel
            const u32* rs = a.big->rep_start + ((size_t)seg_blk0 + b) * 3;
            rep0 = rs[0]; rep1 = rs[1]; rep2 = rs[2];
        }
        if (r.type != 2) {
            if (r.size > cap - op) { if (t == 0) S.err = ZL_E_DSTSIZE_TOO_SMALL; }
            else if (r.type == 0) ex_copy(bdst, blk, r.size);
            else {
                const u32 v4 = blk[0] * 0x01010101u;
                const uint4 v = make_uint4(v4, v4, v4, v4);
                const u32 body = r.size & ~15u;
                for (u32 i = t * 16; i < body; i += EX_THREADS * 16) __builtin_memcpy(bdst + i, &v, 16);
                if (t < r.size - body) bdst[body + t] = (u8)v4;
            }
            __syncthreads();
            if (S.err) break;
            op += r.size;
            continue;
        }
        const bool lit_rle = r.lit_mode == 1;
        const u32 rle_byte = r.lit_a & 0xFF;
        const u8* lit = (r.lit_mode == 0) ? (blk + r.lit_a) : (lit_base + r.lit_dst);
        const u32 nseq = r.nseq, lit_size = r.lit_size;
        const uint4* sq = q_seq + r.seq_dst;  // {literals before sequence i, output bytes before it, offset}; entry nseq: totals
        const u64 cap_left = cap - op;
        u32 lit_pos = 0, out_pos = 0;
        u64 c_scan = 0, c_lit = 0, c_match = 0, c_batches = 0;
        const bool prof = PROF && a.stats != nullptr && t == 0;
        // offsets that name the history the block started with (ZL2_REP_SYM, see the sequence lanes) get their numbers here
#define ZL2_RESOLVE(v_) (((v_) >> 31) ? (((v_) & 3u) == 0 ? rep0 : (((v_) & 3u) == 1 ? rep1 : rep2)) - (((v_) & 0x7FFFFFFFu) >> 2) : (v_))
        for (u32 b0 = 0; b0 < nseq;) {
            const u64 st0 = prof ? __builtin_amdgcn_s_memtime() : 0;
            const u32 window = (nseq - b0 < EX_MAXSEQ_T) ? (nseq - b0) : EX_MAXSEQ_T;
            if (t == 0) S.bcast[1] = 1;
            for (u32 i = t; i <= window; i += EX_THREADS) {  // one extra entry: the next sequence (or the totals)
                const uint4 e = sq[b0 + i];
                S.b_lit[i] = e.x;
                S.b_out[i] = e.y;
                S.b_off[i] = ZL2_RESOLVE(e.z);
            }
            __syncthreads();
            const u32 batch_start = out_pos;
            // sequence i >= 1 joins the batch while its END stays within EX_CAP of the batch start
            for (u32 i = t; i < window; i += EX_THREADS)
                if (i >= 1 && S.b_out[i + 1] - batch_start <= EX_CAP && S.b_out[i + 1] >= batch_start) atomicMax(&S.bcast[1], i + 1);
            __syncthreads();
            const u32 nb = S.bcast[1];
            const u32 batch_end = S.b_out[nb];
            const u32 tot_l = S.b_lit[nb] - lit_pos, tot_o = batch_end - batch_start;
            const u32 gbase = batch_start & ~((1u << EX_GRAN_LOG) - 1);
            const bool linked = tot_o <= EX_CAP;  // otherwise: one long sequence, copied memory to memory
            const u32 ibase = (u32)((uintptr_t)(bdst + batch_start) & 15) - batch_start;  // img index of block position d: d + ibase
            for (u32 i = t; i < nb; i += EX_THREADS) {
                const u32 my_out = S.b_out[i], my_lit = S.b_lit[i];
                const u32 ll = S.b_lit[i + 1] - my_lit;
                const u32 off = S.b_off[i];
                if (S.b_out[i + 1] < my_out + ll || S.b_lit[i + 1] < my_lit) S.err = ZL_E_CORRUPTION;  // positions never decrease
                if ((u64)my_lit + ll > lit_size) S.err = ZL_E_CORRUPTION;   // zstd.c:44030
                if (off == 0 || (u64)off > op + my_out + ll + dict_len) S.err = ZL_E_CORRUPTION;  // zstd.c:44045 (a dictionary's content lies in front of a large frame's first byte)
            }
            if (batch_end > ZL_BLOCK_MAX || batch_end < batch_start || S.b_out[0] != batch_start || S.b_lit[0] != lit_pos) { if (t == 0) S.err = ZL_E_CORRUPTION; }
            else if ((u64)batch_end > cap_left) { if (t == 0) S.err = ZL_E_DSTSIZE_TOO_SMALL; }
            __syncthreads();
            if (S.err) break;
            const u64 st1 = prof ? __builtin_amdgcn_s_memtime() : 0;
            // ---- owner table: granule g starts at gbase + 32 g
            if (linked) {
                for (u32 i = t; i < nb; i += EX_THREADS) {
                    const u32 lo = S.b_out[i], hi = S.b_out[i + 1];
                    u32 g = (lo - gbase + (1u << EX_GRAN_LOG) - 1) >> EX_GRAN_LOG;
                    if (i == 0) g = 0;
                    for (; gbase + (g << EX_GRAN_LOG) < hi; g++) S.own[g] = (u16)i;
                }
            }
            // ---- pass 1: literal bytes of the batch, aligned dwords of the literal buffer
            if (tot_l) {
                const u32 l_end = lit_pos + tot_l;
                const u32 mis = lit_rle ? 0 : (u32)((uintptr_t)(lit + lit_pos) & 3);
                const i32 ja = (i32)lit_pos - (i32)mis;
                for (i32 j0 = ja + 4 * (i32)t; j0 < (i32)l_end; j0 += 8 * EX_THREADS) {
                    // two dwords per trip so that their latencies overlap
                    const i32 j1 = j0 + 4 * EX_THREADS;
                    const bool two = j1 < (i32)l_end;
                    const u32 word0 = lit_rle ? rle_byte * 0x01010101u : *(const u32*)(lit + j0);
                    const u32 word1 = two ? (lit_rle ? rle_byte * 0x01010101u : *(const u32*)(lit + j1)) : 0u;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if (h == 1 && !two) break;
                        const i32 jj = h ? j1 : j0;
                        const u32 word = h ? word1 : word0;
                        const u32 jf = jj < (i32)lit_pos ? lit_pos : (u32)jj;
                        const u32 jl = (u32)(jj + 4) > l_end ? l_end : (u32)(jj + 4);
                        // last sequence whose literal run starts at or before jf
                        u32 lo = 0, hi = nb;
                        while (hi - lo > 1) {
                            const u32 mid = (lo + hi) >> 1;
                            if (S.b_lit[mid] <= jf) lo = mid; else hi = mid;
                        }
                        u32 k = lo;
                        for (u32 j = jf; j < jl; j++) {
                            while (j >= S.b_lit[k + 1]) k++;
                            const u32 d = S.b_out[k] + (j - S.b_lit[k]);
                            const u8 v = (u8)(word >> (8 * (j - (u32)jj)));
                            if (linked) { S.img[d + ibase] = v; S.link[d - batch_start] = EX_FINAL; }
                            else bdst[d] = v;  // (BIG: the link array starts out all-final, only waiting bytes are written)
                        }
                    }
                }
            }
            zl_sync_global();  // everything this workgroup has written so far is in memory before pass 2a reads it back
            const u64 st2 = prof ? __builtin_amdgcn_s_memtime() : 0;
            if (linked) {
                // ---- pass 2a: one hop for every match byte
                const u32 w_first = batch_start >> 2, w_end = (batch_end + 3) >> 2;
                for (u32 w = w_first + t; w < w_end; w += EX_THREADS) {
                    const u32 ws = w << 2;
                    u32 k = owner_of(S, ws < batch_start ? batch_start : ws, gbase);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u32 p = ws + j;
                        if (p < batch_start || p >= batch_end) continue;
                        while (p >= S.b_out[k + 1]) k++;
                        const u32 ll = S.b_lit[k + 1] - S.b_lit[k];
                        const u32 mstart = S.b_out[k] + ll;
                        if (p < mstart) continue;  // literal, placed by pass 1
                        const u32 m = p - mstart, o = S.b_off[k];
                        const i64 q = (m < o) ? (i64)p - (i64)o : (i64)mstart - (i64)o + (i64)(m % o);
                        if (q < (i64)batch_start) {
                            u32 defer = ZL3_ROOT;
                            const u8* from = bdst + q;  // an earlier batch or an earlier block: already in memory
                            if (BIG) {
                                const i64 fpi = (i64)op + q;  // frame position of the source
                                if (fpi < 0) from = dict_end + fpi;  // in front of the frame: the dictionary's content
                                else { const u64 fp = (u64)fpi; defer = fp < seg_start ? (u32)fp : ex_link_load(glink + fp); }  // (another segment | this segment: final, or waiting itself)
                            } else if (dict_len) {
                                const i64 fpi = (i64)op + q;
                                if (fpi < 0) from = dict_end + fpi;
                            }
                            if (defer == ZL3_ROOT) {
                                S.img[p + ibase] = *from;
                                S.link[p - batch_start] = EX_FINAL;
                            } else {
                                S.link[p - batch_start] = (link_t)(defer + EX_CAP);  // waits for frame position `defer` (BIG only)
                            }
                        } else {
                            const u32 qq = (u32)q;
                            const u32 k2 = owner_of(S, qq, gbase);
                            if (qq - S.b_out[k2] < S.b_lit[k2 + 1] - S.b_lit[k2]) {  // literal of this batch: already in place
                                S.img[p + ibase] = S.img[qq + ibase];
                                S.link[p - batch_start] = EX_FINAL;
                            } else {
                                S.link[p - batch_start] = (link_t)(qq - batch_start);
                            }
                        }
                    }
                }
                // ---- pass 2b: pointer jumping over the 16-bit links until nothing is pending.  One barrier
                // per round: the "somebody is still pending" flag rotates over three words.
                if (t == 0) { S.bcast[4] = 0; S.bcast[5] = 0; S.bcast[6] = 0; }
                __syncthreads();
                const u32 ib = batch_start + ibase;  // img index of the batch's first byte
                for (u32 round = 0;; round++) {
                    const u32 fl = 4 + round % 3;
                    u32 pending = 0;
                    for (u32 i = t; i < tot_o; i += EX_THREADS) {
                        const u32 l1 = S.link[i];
                        if (l1 >= EX_CAP) continue;  // final, or waiting for another run
                        const u32 l2 = S.link[l1];
                        if (BIG && l2 >= EX_CAP && l2 != EX_FINAL) {
                            S.link[i] = (link_t)l2;  // what it copies is waiting: it waits for the same frame position
                        } else if (l2 == EX_FINAL) {
                            // (order matters against the other waves of this round: read the link before the byte, write the byte before the link)
                            asm volatile("" ::: "memory");
                            S.img[ib + i] = S.img[ib + l1];
                            asm volatile("" ::: "memory");
                            S.link[i] = EX_FINAL;
                        } else {
                            S.link[i] = (link_t)l2;
                            pending = 1;
                        }
                    }
                    if (pending) S.bcast[fl] = 1;
                    if (t == 0) S.bcast[4 + (round + 1) % 3] = 0;  // next round's flag: last read two barriers ago
                    __syncthreads();
                    if (!S.bcast[fl]) break;
                }
                // ---- flush: img and destination agree modulo 16
                {
                    u8* g = bdst + batch_start;
                    const u32 head = (16 - (ib & 15)) & 15, h = head < tot_o ? head : tot_o;
                    if (t < h) g[t] = S.img[ib + t];
                    const u32 body = (tot_o - h) >> 4;
                    for (u32 i = t; i < body; i += EX_THREADS) {
                        const u32 o = h + (i << 4);
                        *(uint4*)(g + o) = *(const uint4*)(S.img + ib + o);
                    }
                    const u32 tail0 = h + (body << 4);
                    if (tail0 + t < tot_o) g[tail0 + t] = S.img[ib + tail0 + t];
                    if (BIG) {  // what each byte of the batch is: final, or waiting for a frame position
                        u32* gl = glink + op + batch_start;
                        for (u32 i = t; i < tot_o; i += EX_THREADS) {
                            const u32 lk = S.link[i];
                            if (lk != EX_FINAL) { gl[i] = lk - EX_CAP; a.big->tile_flag[(op + batch_start + i) >> ZL3_TILE_LOG] = 1; }
                        }
                    }
                }
            } else {
                // one sequence longer than the image: its literals are in memory (pass 1); every match
                // byte m copies the byte at (match start - offset + m mod offset), which lies in front of the match
                const u32 ll = S.b_lit[1] - S.b_lit[0];
                const u32 mstart = S.b_out[0] + ll, ml = batch_end - mstart, o = S.b_off[0];
                const u8* from = bdst + mstart - o;
                u8* to = bdst + mstart;
                if (BIG) {
                    // byte by byte: a source in front of the segment, or one that is itself waiting, defers the byte
                    const u64 fm = op + mstart;  // frame position of the match
                    for (u32 m = t; m < ml; m += EX_THREADS) {
                        const i64 fpi = (i64)fm - (i64)o + (i64)(m % o);
                        if (fpi < 0) { to[m] = dict_end[fpi]; continue; }  // the dictionary's content
                        const u64 fp = (u64)fpi;
                        const u32 defer = fp < seg_start ? (u32)fp : ex_link_load(glink + fp);
                        if (defer == ZL3_ROOT) to[m] = dst[fp];
                        else { a.big->tile_flag[(fm + m) >> ZL3_TILE_LOG] = 1; glink[fm + m] = defer; }
                    }
                } else if (dict_len && (i64)op + (i64)mstart - (i64)o < 0) {
                    // (a frame that starts from a dictionary, a match longer than a batch whose source begins inside the
                    // dictionary's content: byte by byte, the bytes in front of position 0 from the dictionary)
                    const i64 fm = (i64)op + (i64)mstart;
                    for (u32 m = t; m < ml; m += EX_THREADS) {
                        const i64 fpi = fm - (i64)o + (i64)(m % o);
                        to[m] = fpi < 0 ? dict_end[fpi] : dst[fpi];
                    }
                } else if (o >= 16) {
                    // whole 16-byte pieces while they do not wrap around the period
                    for (u32 m = t * 16; m < ml; m += EX_THREADS * 16) {
                        const u32 s = m % o;
                        if (s + 16 <= o && m + 16 <= ml) {
                            uint4 v;
                            __builtin_memcpy(&v, from + s, 16);
                            __builtin_memcpy(to + m, &v, 16);
                        } else {
                            for (u32 x = m; x < m + 16 && x < ml; x++) to[x] = from[x % o];
                        }
                    }
                } else {
                    for (u32 m = t; m < ml; m += EX_THREADS) to[m] = from[m % o];
                }
            }
            lit_pos += tot_l;
            out_pos += tot_o;
            b0 += nb;
            __syncthreads();  // (LDS reuse; the flush is drained in front of the next pass 2a)
            if (prof) {
                const u64 st3 = __builtin_amdgcn_s_memtime();
                c_scan += st1 - st0; c_lit += st2 - st1; c_match += st3 - st2; c_batches++;
            }
        }
        if (S.err) break;
        const u64 st4 = prof ? __builtin_amdgcn_s_memtime() : 0;
        // trailing literals (zstd.c:44693-44700)
        const u32 rest = lit_size - lit_pos;
        if (lit_pos > lit_size || out_pos + rest > ZL_BLOCK_MAX) { if (t == 0) S.err = ZL_E_CORRUPTION; }
        else if ((u64)out_pos + rest > cap_left) { if (t == 0) S.err = ZL_E_DSTSIZE_TOO_SMALL; }
        __syncthreads();
        if (S.err) break;
        if (lit_rle) { for (u32 i = t; i < rest; i += EX_THREADS) bdst[out_pos + i] = (u8)rle_byte; }
        else ex_copy(bdst + out_pos, lit + lit_pos, rest);
        if (nseq) {  // history after this block (zstd.c:44686)
            const uint4 e = sq[nseq + 1];
            const u32 n0 = ZL2_RESOLVE(e.x), n1 = ZL2_RESOLVE(e.y), n2 = ZL2_RESOLVE(e.z);
            rep0 = n0; rep1 = n1; rep2 = n2;
        }
#undef ZL2_RESOLVE
        __syncthreads();
        const u32 produced = out_pos + rest;
        if (prof) {
            const u64 st5 = __builtin_amdgcn_s_memtime();
            atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)c_scan);
            atomicAdd((unsigned long long*)&a.stats[1], (unsigned long long)c_lit);
            atomicAdd((unsigned long long*)&a.stats[2], (unsigned long long)c_match);
            atomicAdd((unsigned long long*)&a.stats[3], (unsigned long long)(st5 - st4));
            atomicAdd((unsigned long long*)&a.stats[4], (unsigned long long)c_batches);
            atomicAdd((unsigned long long*)&a.stats[5], 1ull);
        }
        op += produced;
    }
    const u64 wg_t2 = wg_prof ? __builtin_amdgcn_s_memtime() : 0;
    zl_sync_global();
    if (wg_prof) {  // profile: prologue, and the final drain of this frame's stores
        atomicAdd((unsigned long long*)&a.stats[-2], (unsigned long long)(wg_t1 - wg_t0));
        atomicAdd((unsigned long long*)&a.stats[-1], (unsigned long long)(__builtin_amdgcn_s_memtime() - wg_t2));
    }
    if (BIG) {  // content size and checksum belong to the whole frame: zl3_finish_kernel
        if (t == 0 && S.err) atomicCAS(&a.big->info->err, 0u, (u32)S.err);
        return S.err == 0;
    }
    if (!S.err) {
        if (t == 0 && fr->fcs != ZL_CONTENTSIZE_UNKNOWN && fr->fcs != op) S.err = ZL_E_CORRUPTION;  // zstd.c:41646
        __syncthreads();
        if (!S.err && fr->has_checksum && !a.ignore_checksum) {
            const u32 got = zl_xxh64_low32(S.img, S.xxh_acc, S.bcast, dst, op);
            if (t == 0 && got != zl_rd_le(src + fr->end_pos, 4)) S.err = ZL_E_CHECKSUM_WRONG;     // zstd.c:41655
        }
        __syncthreads();
    }
    if (t == 0) a.results[f] = S.err ? ZL_ERR(S.err) : op;
    if (wg_prof) atomicAdd((unsigned long long*)&a.stats[-3], (unsigned long long)(__builtin_amdgcn_s_memtime() - wg_t0));  // profile: workgroup lifetime
    return true;
}

template <bool BIG, bool PROF>
__global__ __launch_bounds__(EX_THREADS, BIG ? 6 : 8) void zl2_exec_kernel(zl2_args a) {
    __shared__ __align__(16) ExecSharedT<BIG> S;
    ZL_POISON(S, 2);
    u64 seg_start = 0;
    if (!BIG) {
        u32 g0, g1; bool split;
        zl2_part_range(a, g0, g1, split);
        const u32 slot = g0 * LIT_FRAMES + blockIdx.x;
        if (slot >= g1 * LIT_FRAMES || slot >= a.nframes) return;
        (void)exec_frame<BIG, PROF>(a, S, a.order[slot], seg_start, true);  // long frames first
    } else {
        // a run of consecutive segments per workgroup: matches that stay inside the run are copied here
        // and now, only what reaches in front of it is left to the resolve rounds
        const u32 run = a.big->exec_run;
        const u32 f0 = blockIdx.x * run, f1 = f0 + run < a.nframes ? f0 + run : a.nframes;
        for (u32 f = f0; f < f1; f++) {
            if (!exec_frame<BIG, PROF>(a, S, f, seg_start, f == f0)) break;
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
// The entropy kernels in one role of the schedule.  Grids are sized for the whole batch: which groups a role covers is known on
// the device only; workgroups beyond them return at once.  (The sequence lanes of multi-block launches run on `seq_stream`
// next to the literal lanes and are joined back before this returns.)
static hipError_t zl2_launch_entropy(zl2_args a, u32 role, hipStream_t stream, hipStream_t seq_stream, hipEvent_t fork, hipEvent_t join) {
    const u32 n = a.nframes, ng = (n + LIT_FRAMES - 1) / LIT_FRAMES;
    a.part = role;
    if (seq_stream != stream && (hipEventRecord(fork, stream) != hipSuccess || hipStreamWaitEvent(seq_stream, fork, 0) != hipSuccess)) return hipErrorUnknown;
    hipLaunchKernelGGL(zl2_seq_kernel, dim3(ng), dim3(64), 0, seq_stream, a);
    hipLaunchKernelGGL(zl2_lit_kernel<ZL2_MAXB>, dim3(ng), dim3(LIT_FRAMES * ZL2_MAXB * 4), ZL_LIT_PAD_BYTES, stream, a);
    hipLaunchKernelGGL(zl2_lit_kernel<1>, dim3(ng), dim3(LIT_FRAMES * 4), 0, stream, a);
    hipLaunchKernelGGL(zl2_entropy1_kernel, dim3(3 * ng), dim3(128), 0, stream, a);
    if (seq_stream != stream && (hipEventRecord(join, seq_stream) != hipSuccess || hipStreamWaitEvent(stream, join, 0) != hipSuccess)) return hipErrorUnknown;
    return hipGetLastError();
}
static hipError_t zl2_launch_exec(zl2_args a, u32 role, hipStream_t stream) {
    a.part = role;
    if (a.stats) hipLaunchKernelGGL((zl2_exec_kernel<false, true>), dim3(a.nframes), dim3(EX_THREADS), 0, stream, a);
    else hipLaunchKernelGGL((zl2_exec_kernel<false, false>), dim3(a.nframes), dim3(EX_THREADS), 0, stream, a);
    return hipGetLastError();
}
// A large batch (ZL2_SPLIT_MIN frames or more) may be scheduled in two parts; zl2_order_kernel picks the mode on the device,
// the host always issues the same launches and those that have nothing to do in the chosen mode return at once.
//   Single-block frames (ZL2_MODE_HALVES).  The entropy stage of a frame is one serial chain per block, a CU holds three
//   entropy workgroups, and the exec kernel used to wait for the last of them.  The cheap half of the cost-sorted order now
//   runs entropy AND exec on a second stream next to the expensive half's chains, which leaves only the expensive half's
//   exec behind them (10 000 reference-made 64 KiB frames: 7.0 -> 6.3 ms; three or four parts: much slower, the later ones
//   starve).
//   Multi-block frames (this encoder's own) are not cut.  Measured: halves side by side only get in each other's way --
//   literal and sequence lanes are about level there and every kind of input is as long as the next (decode 3.76 ->
//   4.06 ms); a pipeline over alternate groups (entropy of the odd groups while the exec kernel of the even ones runs)
//   starves the literal workgroups, which need 113 KiB of a CU's LDS to themselves and find exec workgroups sitting there
//   (entropy 1.96 -> 3.80 ms, decode 3.84 -> 4.80 ms).  Even an EMPTY launch of that kernel on the critical path costs: it
//   is not dispatched before a CU has the LDS free (seen as 5.7 instead of 2.6 ms in front of the exec kernel).
// ---------------------------------------------------------------------------------------
// zl2_small_kernel (round 6): SMALL frames that ride on a dictionary's tables, a WAVE each -- config C4's shape (100 000 frames of a few
// hundred bytes).  The lane pipeline gives such a frame a sequence lane, four literal lanes and an exec workgroup of 512 threads, and every
// group of eight frames copies the dictionary's tables (9 KiB in the lanes' formats) into its LDS per frame: 2.5 of the 2.9 ms a call of
// 100 000 objects spends in kernels are those fixed costs.  Here four frames share a workgroup and ONE copy of the tables (zl2_dict_pre);
// a wave stages its frame's block in LDS, decodes the literals (raw, RLE, or one Huffman stream on the dictionary's table: symbol after
// symbol, every lane in step, lane 0 storing), then the sequences on the dictionary's three FSE tables (Repeat_Mode), executing each into
// an LDS image of the output as it is decoded -- literals and match bytes copied by all lanes, a source in front of the frame read from the
// dictionary's content --, and writes the image out in one piece.
// It only ever ACCEPTS: a frame it cannot take (several blocks, a checksum, its own tables, more than 2 KiB of content, four streams) or
// that fails ANY check on the way (a stream not consumed to its last bit, an offset too far, a size that does not add up) is left exactly
// as the index kernel described it, and the lanes decode it -- or name its error -- as before.  Same verdicts as the oracle by construction.
// ---------------------------------------------------------------------------------------
#define SMALLD_FRAMES 8    // frames per workgroup ...
#define SMALLD_G 16        // ... of SMALLD_G lanes each: FOUR frames share a wave.  A frame's chains are serial -- a wave spends the same instructions on them whether 64 lanes follow or 16 --
                           // and only the copies use the lanes: with a wave per frame the kernel was bound by instruction issue (100 000 frames: 1.0 ms)
// A backward bit reader over a stream that lies in LDS: 64 bits of it in registers -- bits [lo, lo + 64), lo a multiple of 32 --, loaded as
// aligned dwords and funnel-shifted (the stream starts at any byte of its buffer); bits in front of the stream's first read as zero, what lies
// behind its last bit is never asked for.  get(bp, n): the n <= 32 bits below position bp.
struct SmallBits {
    const u32* words;   // the dword that holds the stream's first byte
    u32 sh;             // ... and that byte's place in it (0 .. 3)
    i32 lo;
    u64 w;
    __device__ __forceinline__ void open(const u8* buf_aligned16, u32 byte_off) { words = (const u32*)buf_aligned16 + (byte_off >> 2); sh = byte_off & 3; lo = 0x40000000; w = 0; }
    __device__ __forceinline__ void fill(i32 bp) {
        lo = (bp & ~31) - 32;
        const i32 k = lo >> 5;  // the stream's dword index of bit lo (negative: in front of the stream)
        const u32 d0 = k >= 0 ? words[k] : 0u, d1 = k + 1 >= 0 ? words[k + 1] : 0u, d2 = k + 2 >= 0 ? words[k + 2] : 0u;
        w = (u64)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
        if (k < 0 && sh) w &= k + 1 < 0 ? 0ull : ~0xFFFFFFFFull;  // (the dword in front of the stream's first is not the stream's: with a byte shift it would leak in)
    }
    __device__ __forceinline__ u32 get(i32 bp, u32 n) {
        if (n == 0) return 0;
        if (bp - (i32)n < lo || bp > lo + 64) fill(bp);
        return (u32)(w >> (u32)(bp - (i32)n - lo)) & (u32)((1ull << n) - 1);
    }
};
#define SMALLD_BLK 752u    // bytes of the frame's block, of its literals and of its content the wave keeps in LDS: sized for objects of a few hundred bytes, so that eight
#define SMALLD_LIT 496u    // workgroups share a CU (18.6 KiB each); larger frames take the lanes
#define SMALLD_OUT 1008u
struct SmallDecShared {
    u16 tab[2048];
    u32 ent[1280];
    u32 ll_tab[36], ml_tab[53];
    alignas(16) u8 blk[SMALLD_FRAMES][SMALLD_BLK + 16];
    alignas(16) u8 lit[SMALLD_FRAMES][SMALLD_LIT + 16];
    alignas(16) u8 out[SMALLD_FRAMES][SMALLD_OUT + 16];
};
__global__ __launch_bounds__(SMALLD_G * SMALLD_FRAMES) void zl2_small_kernel(zl2_args a) {
    __shared__ SmallDecShared S;
    const u32 t = threadIdx.x, lane = t & (SMALLD_G - 1), wave = t / SMALLD_G;  // ("lane": the lane inside its frame's group; "wave": the group's slot in the workgroup)
    const zl2_dict_pre* const P = a.dict_pre;
    for (u32 i = t; i < 2048; i += SMALLD_G * SMALLD_FRAMES) S.tab[i] = P->tab[i];
    for (u32 i = t; i < 1280; i += SMALLD_G * SMALLD_FRAMES) S.ent[i] = P->ent[i];
    if (t < 36) S.ll_tab[t] = k_LL_base[t] | ((u32)k_LL_bits[t] << 24);
    if (t < 53) S.ml_tab[t] = k_ML_base[t] | ((u32)k_ML_bits[t] << 24);
    __syncthreads();
    // (one wave's LDS traffic is served in the order it was issued: what must not happen is the compiler moving it.  A workgroup-scope fence would
    // also wait for every global access in flight)
    auto lds_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    const u32 f = blockIdx.x * SMALLD_FRAMES + wave;
    if (f >= a.nframes) return;
    zl2_frame_rec* const fr = a.frec + f;
    if (fr->err != 0 || fr->skippable || fr->nblocks != 1 || fr->has_checksum) return;
    const u64 fcs = fr->fcs;
    const zl_frame_desc fd = a.frames[f];
    if (fcs == ZL_CONTENTSIZE_UNKNOWN || fcs > SMALLD_OUT || fcs > fd.dst_cap) return;
    const zl2_blk_rec r = a.brec[(size_t)f * ZL2_MAXB];
    const u8* const src = a.src + fd.src_off + r.src_off;
    u8* const dst = a.dst + fd.dst_off;
    u8* const ob = S.out[wave];
    u32 produced = 0;
    if (r.type == 0) {  // a raw block
        if (r.size != fcs) return;
        for (u32 i = lane; i < r.size; i += SMALLD_G) dst[i] = src[i];
        produced = r.size;
    } else if (r.type == 1) {  // a run of one byte
        if (r.size != fcs) return;
        const u8 b = src[0];
        for (u32 i = lane; i < r.size; i += SMALLD_G) dst[i] = b;
        produced = r.size;
    } else {
        if (r.size > SMALLD_BLK || r.lit_size > SMALLD_LIT || r.lit_mode == 2 || r.nseq > 1023) return;
        if (r.lit_mode == 3 && (P->huf_ok != 1 || r.streams != 1)) return;
        if (r.nseq && (r.seq_modes != 0xFCu || P->seq_ok != 1)) return;
        u8* const blk = S.blk[wave];
        for (u32 i = 16 * lane; i < r.size; i += 16 * SMALLD_G) {
            if (i + 16 <= r.size) { uint4 x; __builtin_memcpy(&x, src + i, 16); *(uint4*)(blk + i) = x; }
            else for (u32 b = i; b < r.size; b++) blk[b] = src[b];
        }
        u8* const lit = S.lit[wave];
        lds_sync();
        // ---- literals
        const u32 nlit = r.lit_size;
        if (r.lit_mode == 0) { for (u32 i = lane; i < nlit; i += SMALLD_G) lit[i] = blk[r.lit_a + i]; }
        else if (r.lit_mode == 1) { for (u32 i = lane; i < nlit; i += SMALLD_G) lit[i] = (u8)r.lit_a; }
        else {
            // one Huffman stream, read from its end (zstd.c:38653): the table's `log` top bits name a cell {symbol, length}
            const u8* const hs = blk + r.lit_a;
            const u32 hn = r.lit_b, log = P->huf_log;
            i32 bp = zl_bwd_init(hs, hn);
            if (bp < 0 || log == 0 || log > 11) return;
            SmallBits rd;
            rd.open(blk, r.lit_a);
            for (u32 i = 0; i < nlit; i++) {
                const u32 cell = S.tab[rd.get(bp, log)];
                bp -= (i32)(cell >> 8);
                if (lane == 0) lit[i] = (u8)cell;
                if (bp < 0) return;  // (uniform) ran out of bits
            }
            if (bp != 0) return;  // every stream is consumed exactly (the oracle's rule)
        }
        lds_sync();
        // ---- sequences, executed as they are decoded
        u32 lit_used = 0, pos = 0;
        if (r.nseq) {
            const u8* const bs = blk + r.seq_off + 1;  // (behind the mode byte: three times Repeat_Mode)
            if (r.seq_len < 1) return;
            const u32 bn = r.seq_len - 1;
            i32 bp = zl_bwd_init(bs, bn);
            if (bp < 0) return;
            SmallBits rd;
            rd.open(blk, r.seq_off + 1);
            const u32 ll_log = P->seq_log[0], of_log = P->seq_log[1], ml_log = P->seq_log[2];
            u32 sl = rd.get(bp, ll_log); bp -= (i32)ll_log;
            u32 so = rd.get(bp, of_log); bp -= (i32)of_log;
            u32 sm = rd.get(bp, ml_log); bp -= (i32)ml_log;
            if (bp < 0) return;
            u32 r0 = a.dict_rep[0], r1 = a.dict_rep[1], r2 = a.dict_rep[2];
            const u32 dict_len = a.dict_len;
            for (u32 i = 0; i < r.nseq; i++) {
                const u32 cl = S.ent[sl], cm = S.ent[512 + sm], co = S.ent[1024 + so];
                const u32 yl = (cl >> 16) & 0xFF, ym = (cm >> 16) & 0xFF, yo = (co >> 16) & 0xFF;
                if (yl > ZL_MAX_LL || ym > ZL_MAX_ML || yo > ZL_MAX_OF) return;
                const u32 tl = S.ll_tab[yl], tm = S.ml_tab[ym];
                const u32 nl = tl >> 24, nm = tm >> 24;
                const u32 ofv = (1u << yo) + rd.get(bp, yo); bp -= (i32)yo;
                const u32 ml = (tm & 0xFFFFFF) + rd.get(bp, nm); bp -= (i32)nm;
                const u32 ll = (tl & 0xFFFFFF) + rd.get(bp, nl); bp -= (i32)nl;
                if (bp < 0) return;
                u32 off;
                if (ofv > 3) { off = ofv - 3; r2 = r1; r1 = r0; r0 = off; }
                else {
                    const u32 idx = ofv - 1 + (ll == 0);
                    if (idx == 0) off = r0;
                    else {
                        off = (idx == 3) ? r0 - 1 : (idx == 1 ? r1 : r2);
                        if (off == 0) return;
                        if (idx != 1) r2 = r1;
                        r1 = r0;
                        r0 = off;
                    }
                }
                if (lit_used + ll > nlit || (u64)pos + ll + ml > fcs || off > pos + ll + dict_len) return;
                // the literals, then the match: its source is in front of it -- in the image, or (in front of position 0) in the dictionary's content;
                // a source that overlaps the match repeats its first `off` bytes
                for (u32 j = lane; j < ll; j += SMALLD_G) ob[pos + j] = lit[lit_used + j];
                pos += ll; lit_used += ll;
                lds_sync();
                for (u32 j = lane; j < ml; j += SMALLD_G) {
                    const u32 k = off >= ml ? j : j % off;     // (byte j copies byte j - off, which copied j - 2 off ...: the first `off` bytes, round and round)
                    const i32 at = (i32)pos - (i32)off + (i32)k;
                    ob[pos + j] = at >= 0 ? ob[at] : a.dict_end[at];  // (the dictionary's last 8 KiB kept in LDS instead: measured SLOWER, 1.35 -> 1.64 ms per 100 000 frames -- a copy per workgroup and a fifth fewer of them per CU; the reads hit the caches and other waves cover them)
                }
                pos += ml;
                lds_sync();
                if (i + 1 < r.nseq) {
                    u32 nb = (cl >> 10) & 0xF;
                    sl = (cl & 0x3FF) + rd.get(bp, nb); bp -= (i32)nb;
                    nb = (cm >> 10) & 0xF;
                    sm = (cm & 0x3FF) + rd.get(bp, nb); bp -= (i32)nb;
                    nb = (co >> 10) & 0xF;
                    so = (co & 0x3FF) + rd.get(bp, nb); bp -= (i32)nb;
                    if (bp < 0) return;
                }
            }
            if (bp != 0) return;  // zstd.c:44690
        }
        // the literals behind the last sequence
        const u32 rest = nlit - lit_used;
        if ((u64)pos + rest != fcs) return;
        for (u32 j = lane; j < rest; j += SMALLD_G) ob[pos + j] = lit[lit_used + j];
        pos += rest;
        lds_sync();
        for (u32 i = 16 * lane; i < pos; i += 16 * SMALLD_G) {
            if (i + 16 <= pos) { const uint4 x = *(const uint4*)(ob + i); __builtin_memcpy(dst + i, &x, 16); }
            else for (u32 b = i; b < pos; b++) dst[b] = ob[b];
        }
        produced = pos;
    }
    if (lane == 0) {
        a.results[f] = produced;
        fr->cost = 0; fr->lit_cost = 0;
        fr->err = ZL_E_DONE;  // (the lanes and the exec kernel leave such a frame alone: zl2_index_kernel's word for "decoded before this launch")
        atomicAdd(a.max_blocks + 3, 1u);  // (a batch this kernel finished on its own: zl2_order_kernel calls the lanes off)
    }
}

extern "C" hipError_t zl2_launch(const zl2_args* a, hipStream_t stream, const zl2_side* side, hipEvent_t* ev /* optional: 4 events, recorded after index / entropy / exec */) {
    if (a->nframes == 0) return hipSuccess;
    const u32 n = a->nframes;
    hipLaunchKernelGGL(zl2_index_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, *a);
    if (ZL2_SMALL && a->dict_pre != nullptr && a->dict_end != nullptr && a->big == nullptr)  // small frames on a dictionary's tables: a wave each, in front of the lanes
        hipLaunchKernelGGL(zl2_small_kernel, dim3((n + SMALLD_FRAMES - 1) / SMALLD_FRAMES), dim3(SMALLD_G * SMALLD_FRAMES), 0, stream, *a);
    hipLaunchKernelGGL(zl2_order_kernel, dim3(1), dim3(1024), 0, stream, *a);
    if (ev) (void)hipEventRecord(ev[0], stream);
    {   // the literal kernel's dynamic LDS has to be allowed once per DEVICE (one process may drive several: ZL_d_devices)
        static std::atomic<unsigned long long> shaped{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(shaped.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)zl2_lit_kernel<ZL2_MAXB>, hipFuncAttributeMaxDynamicSharedMemorySize, ZL_LIT_PAD_BYTES) != hipSuccess) return hipErrorUnknown;
            shaped.fetch_or(bit, std::memory_order_release);
        }
    }
    hipError_t e;
    const bool two = side->part_stream[0] != nullptr && n >= ZL2_SPLIT_MIN;
    hipStream_t s2 = side->part_stream[0];
    // the second stream waits for the order; the main stream's part is launched first (its workgroups take their CUs first)
    if (two && (hipEventRecord(side->fork, stream) != hipSuccess || hipStreamWaitEvent(s2, side->fork, 0) != hipSuccess)) return hipErrorUnknown;
    if ((e = zl2_launch_entropy(*a, two ? ZL2_ROLE_E1 : 0u, stream, side->stream, side->fork, side->join)) != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[1], stream);
    if ((e = zl2_launch_exec(*a, two ? ZL2_ROLE_X1 : 0u, stream)) != hipSuccess) return e;
    if (two) {
        if ((e = zl2_launch_entropy(*a, ZL2_ROLE_ES, s2, s2, nullptr, nullptr)) != hipSuccess) return e;
        if ((e = zl2_launch_exec(*a, ZL2_ROLE_XS, s2)) != hipSuccess) return e;
        if (hipEventRecord(side->part_join[0], s2) != hipSuccess || hipStreamWaitEvent(stream, side->part_join[0], 0) != hipSuccess) return hipErrorUnknown;
    }
    if (ev) (void)hipEventRecord(ev[2], stream);
    return hipGetLastError();
}

extern "C" hipError_t zl2_launch_dict_prebuild(const zl2_blk_rec* ext, const u8* ext_base, zl2_dict_pre* out, hipStream_t stream) {
    hipLaunchKernelGGL(zl2_dict_prebuild_kernel, dim3(1), dim3(64), 0, stream, ext, ext_base, out);
    return hipGetLastError();
}
// Large-frame pipeline (zl_decode3.hip): the same kernels over the segments of one frame.  The caller has
// filled frec / brec / plan / ext on the device.
extern "C" hipError_t zl2_launch_segments_entropy(const zl2_args* a, hipStream_t stream, const zl2_side* side) {
    const u32 n = a->nframes;  // (the caller has set part = 0: the whole order)
    hipLaunchKernelGGL(zl2_order_kernel, dim3(1), dim3(1024), 0, stream, *a);
    const u32 ngroups = (n + LIT_FRAMES - 1) / LIT_FRAMES;
    if (hipEventRecord(side->fork, stream) != hipSuccess || hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess) return hipErrorUnknown;
    hipLaunchKernelGGL(zl2_seq_kernel, dim3(ngroups), dim3(64), 0, side->stream, *a);
    if (hipFuncSetAttribute((const void*)zl2_lit_kernel<ZL2_MAXB>, hipFuncAttributeMaxDynamicSharedMemorySize, ZL_LIT_PAD_BYTES) != hipSuccess) return hipErrorUnknown;
    hipLaunchKernelGGL(zl2_lit_kernel<ZL2_MAXB>, dim3(ngroups), dim3(LIT_FRAMES * ZL2_MAXB * 4), ZL_LIT_PAD_BYTES, stream, *a);
    hipLaunchKernelGGL(zl2_lit_kernel<1>, dim3(ngroups), dim3(LIT_FRAMES * 4), 0, stream, *a);
    hipLaunchKernelGGL(zl2_entropy1_kernel, dim3(3 * ngroups), dim3(128), 0, stream, *a);
    if (hipEventRecord(side->join, side->stream) != hipSuccess || hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) return hipErrorUnknown;
    return hipGetLastError();
}
extern "C" hipError_t zl2_launch_segments_exec(const zl2_args* a, u32 exec_run, hipStream_t stream) {
    hipLaunchKernelGGL((zl2_exec_kernel<true, false>), dim3((a->nframes + exec_run - 1) / exec_run), dim3(EX_THREADS), 0, stream, *a);
    return hipGetLastError();
}

// debug: what the runtime says about resident workgroups per CU (not part of include/zl_hip.h)
extern "C" int zl2_debug_occupancy(int which) {
    int n = -1;
    if (which == 0) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zl2_exec_kernel<false, false>, EX_THREADS, 0);
    else if (which == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zl2_lit_kernel<ZL2_MAXB>, LIT_FRAMES * ZL2_MAXB * 4, ZL_LIT_PAD_BYTES);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zl2_seq_kernel, 64, 0);
    return n;
}
