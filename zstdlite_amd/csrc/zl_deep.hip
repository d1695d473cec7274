// zl_deep.hip -- the deep match finder behind levels >= 6 for gfx950 (MI355X).
//
// The reference's levels 6-15 are its lazy strategies: at every position they look at the 2^searchLog most recent earlier
// positions with the same hash of minMatch bytes inside a window of 2 - 4 MiB (hash chains and the row hash,
// zstd.c:31546-33746 ZSTD_HcFindBestMatch / ZSTD_RowFindBestMatch), keep the longest, and weigh it against the matches one and
// two positions on (ZSTD_compressBlock_lazy_generic, depth 1 / 2).  A chain is built by walking the input in order; this
// machine cannot do that, but it can SORT:
//
//   zl_deep_key_kernel      every position p of a section (64 MiB) becomes the key  hash(minMatch bytes at p) << 32 | p
//   radix sort              of the keys by their hash bits only (rocPRIM's device radix sort through hipcub: a plain library
//                           primitive, as a GEMM would be).  The sort is stable, so inside one hash value the positions ascend:
//                           the K entries in front of a position in the sorted order ARE its K most recent earlier occurrences
//                           -- exactly the candidates a hash chain of depth K visits, whatever order the hardware worked in.
//   zl_deep_search_kernel   a workgroup takes 256 consecutive sorted entries plus the K in front of them, fetches the 64 bytes
//                           [p - 8, p + 56) around every one of these positions ONCE (the only gathers of the whole search:
//                           one record per position instead of one per candidate) and keeps them in LDS; every lane then
//                           measures its K predecessors against its own record -- LDS reads only --, keeps the best by the
//                           lazy parsers' gain (4 * length - log2(offset), zstd.c:31653), counts the bytes that agree in
//                           FRONT of both (the "catch up" of zstd.c:31707) and writes one word for its position.
//   zl_match_deep_kernel    the window-parallel parse of zl_match_auto_kernel (zl_encode.hip) with its tables taken away: a
//                           window's 256 lanes read their words (one coalesced load), try the repeat offset of the window in
//                           front out of an LDS copy of the bytes it points at, apply the one- and two-step lazy rule, close
//                           the greedy path with lane shuffles, extend backwards over free literals and emit.  No gathers,
//                           no table, no inserts: what is left is the parse.
//
// Matches are measured over ZL_DEEP_CAP bytes; a longer one arrives as several pieces with one offset, which the pass behind
// the parse joins (zl_merge_pieces).  CPU model with the reference's entropy stage behind it: scripts/dev/sim/deep_sim.c
// (python sources, one frame: 1.001 / 0.998 / 0.996 of the reference's size at levels 6 / 9 / 12).
#include <hip/hip_runtime.h>
#include <atomic>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "zl_device.h"
#include "zl_encode.h"
#include "zl_format.h"
#include "zl_kernels.h"
#include "zl_match_common.h"

#define DEEP_THREADS 256
#define DEEP_SEC_BYTES ((u64)ZL_CHUNK << ZL_FAR_SECTION_LOG)  // 64 MiB: what is sorted at once, and what no offset crosses
#define DEEP_SEC_MAX ((u32)DEEP_SEC_BYTES)

// ---------------------------------------------------------------------------------------
// keys: hash of the position's minMatch bytes above the position itself
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEEP_THREADS) void zl_deep_key_kernel(const u8* sec, u32 n, u32 mml_shift, u64* keys) {
    const u32 i = blockIdx.x * DEEP_THREADS + threadIdx.x;
    if (i >= n) return;
    u64 v = 0;
    if (i + 8 <= n) v = ld64(sec + i);
    else for (u32 b = 0; i + b < n; b++) v |= (u64)sec[i + b] << (8 * b);  // (the section's last bytes: zero-padded, like their records below)
    const u32 h = (u32)(((v << mml_shift) * 0x9E3779B185EBCA87ULL) >> (64 - ZL_DEEP_HLOG));
    keys[i] = ((u64)h << 32) | i;
}

// 16 bytes at section offset a (may be negative or run past the end): bytes outside [0, n) read as zero
__device__ __forceinline__ uint4 deep_ld16(const u8* sec, i32 a, u32 n) {
    if (a >= 0 && (u32)a + 16 <= n) return ld128(sec + a);
    u32 w[4] = {0, 0, 0, 0};
    for (u32 b = 0; b < 16; b++) {
        const i32 q = a + (i32)b;
        if (q >= 0 && (u32)q < n) w[b >> 2] |= (u32)sec[q] << (8 * (b & 3));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ u32 cb64(u32 a0, u32 a1, u32 b0, u32 b1) {  // equal bytes from the start of two 8-byte words
    return common_bytes((u64)a0 | ((u64)a1 << 32), (u64)b0 | ((u64)b1 << 32));
}
__device__ __forceinline__ u32 cb128(const uint4& a, const uint4& b) {
    const u32 l0 = cb64(a.x, a.y, b.x, b.y);
    return l0 == 8 ? 8 + cb64(a.z, a.w, b.z, b.w) : l0;
}

// ---------------------------------------------------------------------------------------
// search: entry i of the sorted order against entries i - 1 ... i - K
// ---------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(DEEP_THREADS) void zl_deep_search_kernel(const u64* sorted, u32 n, const u8* sec, u32 K, u32 window, u32 min_len, u32 frames, u64* best, u64* chain) {
    constexpr u32 E = DEEP_THREADS + KMAX;
    __shared__ uint4 rec[4][E];  // rec[j][e]: bytes [p - 8 + 16 j, p + 8 + 16 j) of entry e's position (one array per slice: consecutive lanes read consecutive slots)
    __shared__ u64 skey[E + 1];  // hash << 32 | position of every entry (one entry behind the block: does the last one have a twin?); no entry: a hash no position has
    static_assert(ZL_DEEP_HLOG < 32, "0xFFFFFFFF is not a hash");
    const u32 t = threadIdx.x;
    const u32 i0 = blockIdx.x * DEEP_THREADS;
    // positions and hashes first: an entry whose hash neither neighbour shares has no candidate and is nobody's candidate
    for (u32 e = t; e < E + 1; e += DEEP_THREADS) {
        const i64 idx = (i64)i0 + (i64)e - KMAX;
        u64 key = 0xFFFFFFFF00000000ull;
        if (idx >= 0 && idx < (i64)n && (e >= (u32)KMAX || e + K >= (u32)KMAX)) key = sorted[idx];  // (of the entries in front only the K nearest are looked at)
        skey[e] = key;
    }
    __syncthreads();
    for (u32 e = t; e < E; e += DEEP_THREADS) {
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        const u64 key = skey[e];
        const u32 h = (u32)(key >> 32);
        const bool twin = (e > 0 && (u32)(skey[e - 1] >> 32) == h) || (u32)(skey[e + 1] >> 32) == h;
#if defined(ZL_DEEP_EXP) && (ZL_DEEP_EXP & 2)
        if (h != 0xFFFFFFFFu && twin && (u32)key == 0x7FFFFFF1u) {  // (experiment: no record loads)
#else
        if (h != 0xFFFFFFFFu && twin) {
#endif
            const i32 a = (i32)(u32)key - 8;
            r0 = deep_ld16(sec, a, n); r1 = deep_ld16(sec, a + 16, n); r2 = deep_ld16(sec, a + 32, n); r3 = deep_ld16(sec, a + 48, n);
        }
        rec[0][e] = r0; rec[1][e] = r1; rec[2][e] = r2; rec[3][e] = r3;
    }
    __syncthreads();
    const u32 e = t + KMAX;
    if (i0 + t >= n) return;
    const u32 p = (u32)skey[e], hme = (u32)(skey[e] >> 32);
    const uint4 m0 = rec[0][e], m1 = rec[1][e], m2 = rec[2][e], m3 = rec[3][e];
    const u32 chunk0 = p & ~(u32)(ZL_CHUNK - 1);  // (sections start on a chunk boundary)
    // frames: bit 0 -- independent frames (a candidate lies inside the position's own chunk); bit 1 -- a SECOND pass over the section,
    // sorted by a longer hash (levels 16+): what it finds is merged into the chain the first pass left, the best words stay as they are
    const bool merge = (frames & 2u) != 0;
    frames &= 1u;
    u32 lo = p > window ? p - window : 0u;
    if (frames && lo < chunk0) lo = chunk0;
    u32 bl = 0, bo = 0, be = 0;
    int bg = -1000000;
    bool alive = true;
    // levels 16+ (chain != null): every candidate longer than all nearer ones, the ZL_DEEP_CHAIN longest kept -- ascending, the longest last
    u32 ch_l[ZL_DEEP_CHAIN], ch_o[ZL_DEEP_CHAIN];
#pragma unroll
    for (int j = 0; j < ZL_DEEP_CHAIN; j++) { ch_l[j] = 0; ch_o[j] = 0; }
    for (u32 k = 1; k <= K; k++) {
        const u32 ce = e - k;
        const u64 ck = skey[ce];
        const u32 q = (u32)ck;
        alive = alive && (u32)(ck >> 32) == hme && q >= lo && q < p;  // (positions ascend inside a hash value: the first one below the window ends the walk)
        if (!__any(alive)) break;
        if (alive) {
            const uint2 c0 = *(const uint2*)&rec[0][ce].z;
            u32 len = cb64(m0.z, m0.w, c0.x, c0.y);
            if (len == 8) {
                u32 l = cb128(m1, rec[1][ce]);
                len += l;
                if (l == 16) {
                    l = cb128(m2, rec[2][ce]);
                    len += l;
                    if (l == 16) len += cb128(m3, rec[3][ce]);
                }
            }
            if (len >= min_len) {
                const int g = 4 * (int)len - (int)(31 - __clz(p - q + 1));
                if (g > bg) { bg = g; bl = len; bo = p - q; be = ce; }  // (ties: the nearer one, met first)
                if (chain && len > ch_l[ZL_DEEP_CHAIN - 1]) {
#pragma unroll
                    for (int j = 0; j + 1 < ZL_DEEP_CHAIN; j++) { ch_l[j] = ch_l[j + 1]; ch_o[j] = ch_o[j + 1]; }
                    ch_l[ZL_DEEP_CHAIN - 1] = len; ch_o[ZL_DEEP_CHAIN - 1] = p - q;
                }
            }
        }
    }
    // the best candidate's word: its length, the bytes that agree in FRONT of both (the sequence may start there if the parse leaves those literals free), its offset
    auto best_word = [&]() -> u64 {
        const u64 mine = (u64)m0.x | ((u64)m0.y << 32), theirs = (u64)rec[0][be].x | ((u64)rec[0][be].y << 32);
        const u64 x = mine ^ theirs;
        u32 back = x ? (u32)(__builtin_clzll(x) >> 3) : 8u;
        const u32 q = p - bo;
        const u32 lo_src = frames ? chunk0 : 0u;  // the source may not start in front of this (the section, or the frame)
        if (back > q - lo_src) back = q - lo_src;
        if (back > p - chunk0) back = p - chunk0;  // nor the sequence in front of its chunk
        return (u64)(bl | (back << 16)) | ((u64)bo << 32);
    };
    if (merge) {
        // the best word: this pass's best by the same gain against the one the first pass left (whose length may have been measured on since)
        if (bl) {
            const u64 old = best[p];
            const u32 ol = (u32)old & 0xFFFFu, oo = (u32)(old >> 32);
            const int og = ol ? 4 * (int)ol - (int)(31 - __clz(oo + 1)) : -1000000;
            if (bg > og) best[p] = best_word();
        }
        if (chain == nullptr) return;
        // The two chains -- this pass's and the one in memory -- are each ascending in length and descending in offset; together the entries that no
        // other one beats at both (at least as long AND at least as near) are the nearest offset per length over all candidates of both passes.  The
        // ZL_DEEP_CHAIN longest of them stay.
        static_assert(ZL_DEEP_CHAIN == 4, "two 16-byte words each way");
        if (ch_l[ZL_DEEP_CHAIN - 1] == 0) return;  // nothing new
        uint4* const at = (uint4*)(chain + (size_t)ZL_DEEP_CHAIN * p);
        const uint4 e0 = at[0], e1 = at[1];
        u32 L[8] = {e0.x, e0.z, e1.x, e1.z, ch_l[0], ch_l[1], ch_l[2], ch_l[3]};
        u32 O[8] = {e0.y, e0.w, e1.y, e1.w, ch_o[0], ch_o[1], ch_o[2], ch_o[3]};
        bool dead[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            bool d = (L[x] & 0xFFFFu) == 0;
#pragma unroll
            for (int y = 0; y < 8; y++) if (y != x) {
                const u32 lx = L[x] & 0xFFFFu, ly = L[y] & 0xFFFFu;
                // y beats x: at least as long and at least as near (of two equal entries the first stays)
                d = d || (ly != 0 && ly >= lx && O[y] <= O[x] && (ly > lx || O[y] < O[x] || y < x));
            }
            dead[x] = d;
        }
        u32 nl[4] = {0, 0, 0, 0}, no[4] = {0, 0, 0, 0};
#pragma unroll
        for (int slot = 3; slot >= 0; slot--) {  // the longest survivor not yet taken, four times
            u32 bestl = 0, besto = 0; int which = -1;
#pragma unroll
            for (int x = 0; x < 8; x++) { const u32 lx = L[x] & 0xFFFFu; if (!dead[x] && lx > bestl) { bestl = lx; besto = O[x]; which = x; } }
#pragma unroll
            for (int x = 0; x < 8; x++) if (x == which) dead[x] = true;
            nl[slot] = bestl; no[slot] = besto;
        }
        at[0] = make_uint4(nl[0], no[0], nl[1], no[1]);
        at[1] = make_uint4(nl[2], no[2], nl[3], no[3]);
        return;
    }
    u64 word = 0;
    if (bl) word = best_word();
#if defined(ZL_DEEP_EXP) && (ZL_DEEP_EXP & 1)
    if (word == 0x123456789ABCDEFull) best[p] = word;  // (experiment: no scattered store)
#elif defined(ZL_DEEP_EXP) && (ZL_DEEP_EXP & 4)
    best[i0 + t] = word;  // (experiment: the store in sorted order, coalesced)
#else
    best[p] = word;
#endif
    if (chain) {
        static_assert(ZL_DEEP_CHAIN == 4, "two 16-byte stores");
        uint4* const to = (uint4*)(chain + (size_t)ZL_DEEP_CHAIN * p);
        to[0] = make_uint4(ch_l[0], ch_o[0], ch_l[1], ch_o[1]);
        to[1] = make_uint4(ch_l[2], ch_o[2], ch_l[3], ch_o[3]);
    }
}


// ---------------------------------------------------------------------------------------
// extension: a match that filled its record (ZL_DEEP_CAP bytes) is measured on, so that the parse steps over it whole and the
// lazy rule sees what it is worth.  Position order, a wave per 64 positions: position p with the offset of p - 1, both full, is
// the INTERIOR of one longer match (its length is its neighbour's less one); every other full one is a HEAD, and the whole wave
// measures a head's next ZL_DEEP_EXT bytes at once, sixteen per lane.  (python sources, 1 MB of one file, level 9: 1.069 -> 1.03 of
// the reference's size -- scripts/dev/sim/deep_sim.c cap=56 / 255.)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(DEEP_THREADS) void zl_deep_extend_kernel(u64* best, u32 n, const u8* sec, u64* chain) {
    const u32 p = blockIdx.x * DEEP_THREADS + threadIdx.x;
    const u32 lane = threadIdx.x & 63;
    const u64 w = p < n ? best[p] : 0ull;
    const u32 len = (u32)w & 0xFFFFu, off = (u32)(w >> 32);
    const bool full = len == ZL_DEEP_CAP;
    const u64 fm = __ballot(full);
    if (fm == 0) return;  // (uniform)
    const u32 plen = __shfl_up(len, 1, 64), poff = __shfl_up(off, 1, 64);
    const bool head = full && (lane == 0 || plen != ZL_DEEP_CAP || poff != off);
    const u64 heads = __ballot(head);
    u32 mine_ext = 0;
    for (u64 hm = heads; hm; hm &= hm - 1) {
        const u32 h = (u32)__builtin_ctzll(hm);
        const u32 hp = p - lane + h, hoff = __shfl(off, (int)h, 64);
        const u32 a = hp + ZL_DEEP_CAP + 16 * lane;
        u32 part = 0;
        if (a + 16 <= n) part = cb128(ld128(sec + a), ld128(sec + a - hoff));
        else for (; part < 16 && a + part < n && sec[a + part] == sec[a + part - hoff]; part++) {}
        const u64 whole = __ballot(part == 16);
        const u32 first = ~whole ? (u32)__builtin_ctzll(~whole) : 64u;  // the first lane whose sixteen bytes do not all agree
        const u32 tail = __shfl(part, (int)(first & 63), 64);
        if (lane == h) mine_ext = first * 16 + (first < 64 ? tail : 0u);
    }
    // every full position: its head is the nearest one at or below it
    const u64 at_or_below = heads & ((2ull << lane) - 1);
    const u32 h = at_or_below ? 63u - (u32)__builtin_clzll(at_or_below) : 0u;
    const u32 h_ext = __shfl(mine_ext, (int)h, 64);
    if (full) {
        const u32 d = lane - h;
        const u32 total = ZL_DEEP_CAP + (h_ext > d ? h_ext - d : 0u);
        best[p] = (w & ~0xFFFFull) | total;
        if (chain) {  // the chain's entry of the same offset (a full one is the longest or shares its length with it) learns its real length too
            u64* const ch = chain + (size_t)ZL_DEEP_CHAIN * p;
#pragma unroll
            for (int j = 0; j < ZL_DEEP_CHAIN; j++) { const u64 cwj = ch[j]; if ((u32)(cwj >> 32) == off && ((u32)cwj & 0xFFFFu) == ZL_DEEP_CAP) ch[j] = (cwj & ~0xFFFFull) | total; }
        }
    }
}

// ---------------------------------------------------------------------------------------
// the parse
// ---------------------------------------------------------------------------------------
#define DEEP_WBUF 352  // a window's own bytes: 256 positions + ZL_DEEP_CAP + a last word and its alignment slack
#define DEEP_RBUF 368  // the bytes the repeat offset points at, from 8 in front of the window's first source on
struct DeepShared {
    u64 cand[DEEP_THREADS + 2];  // [31:0] length, [63:32] offset of each lane's match (0: none); two zero words behind the window: what the lazy rule reads there
    u16 jump[DEEP_THREADS];
    u32 scan_a[4];
    u32 rep;                     // offset of the most recent sequence
    u32 probe_good;              // cheap-bytes probe: sequences that were long, or used the offset of the one before
    u32 t_end[DEEP_THREADS], t_off[DEEP_THREADS];  // the merge pass
    alignas(16) u8 wbuf[DEEP_WBUF];
    alignas(16) u8 rbuf[DEEP_RBUF];
};
static_assert(255 + ZL_DEEP_CAP + 16 + 4 <= DEEP_WBUF && 8 + 255 + ZL_DEEP_CAP + 16 + 4 <= DEEP_RBUF, "the last lane's last round stays inside the buffers");

// eight bytes at byte offset o of an LDS buffer: three aligned words, funnel-shifted
__device__ __forceinline__ u64 lds64(const u8* buf, u32 o) {
    const u32* const w = (const u32*)buf + (o >> 2);
    const u32 d0 = w[0], d1 = w[1], d2 = w[2], sh = o & 3;
    return (u64)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
}
__device__ __forceinline__ u32 lds_common16(const u8* a, u32 oa, const u8* b, u32 ob) {
    const u32 l0 = common_bytes(lds64(a, oa), lds64(b, ob));
    return l0 == 8 ? 8 + common_bytes(lds64(a, oa + 8), lds64(b, ob + 8)) : l0;
}

__global__ __launch_bounds__(DEEP_THREADS) void zl_match_deep_kernel(zl_match_args a, u32 nchunks) {
    __shared__ DeepShared S;
    const u32 t = threadIdx.x;
    // a.sub > 1 (a launch of few chunks, where a workgroup's walk over its chunk -- 512 windows, one behind the other -- is all the
    // call waits for): `sub` workgroups share a chunk, each parses its part [p_lo, p_hi) as if the chunk began there (no repeat
    // offset, no match running in from the part before; a match ends with the part) and leaves literals, sequences and counts where
    // zl_deep_join_kernel collects them -- which also joins a match cut at a part's end with its continuation (zl_merge_pieces)
    const u32 parts = a.sub > 1 ? a.sub : 1u;
    const u32 part = parts > 1 ? blockIdx.x % parts : 0u;
    const u32 c = parts > 1 ? blockIdx.x / parts : (a.order ? a.order[zl_spread(blockIdx.x, nchunks)] : zl_spread(blockIdx.x, nchunks));
    const zl_chunk_desc cd = a.chunks[c];
    const u8* src = a.src + cd.src_off;
    const u32 n = cd.src_size;
    const u32 p_lo = part * (ZL_CHUNK / parts);
    const u32 p_hi = parts > 1 ? (n < p_lo + ZL_CHUNK / parts ? n : p_lo + ZL_CHUNK / parts) : n;
    if (parts > 1 && p_lo >= n) {
        if (t == 0) { a.sub_meta[2 * blockIdx.x] = 0; a.sub_meta[2 * blockIdx.x + 1] = 0; }
        return;
    }
    u8* lits = parts > 1 ? a.sub_lits + (size_t)c * a.lit_stride + p_lo : a.lits + (size_t)c * a.lit_stride;
    zl_seq* seqs = parts > 1 ? a.sub_seqs + (size_t)c * a.seq_stride + (size_t)part * (a.seq_stride / parts) : a.seqs + (size_t)c * a.seq_stride;
    const u64* const best = a.deep_best + (size_t)(cd.src_off - a.deep_lo);
    // what may be read around the chunk: its section (no offset crosses one, zl_encode.h) and the input's end
    const u64 sec_off = cd.src_off & (DEEP_SEC_BYTES - 1);
    const u32 reach = a.joined ? (u32)sec_off : 0u;                                            // bytes in front of the chunk a source may lie in
    const u64 sec_end = cd.src_off - sec_off + DEEP_SEC_BYTES;
    const u32 avail = (u32)((a.total < sec_end ? a.total : sec_end) - cd.src_off);               // bytes readable from the chunk's first byte on
    auto ld16 = [&](i32 o) -> uint4 {  // 16 bytes at chunk offset o; outside [-reach, avail): zero
        if (o >= -(i32)reach && (i64)o + 16 <= (i64)avail) return ld128(src + o);
        u32 w[4] = {0, 0, 0, 0};
        for (u32 b = 0; b < 16; b++) {
            const i32 q = o + (i32)b;
            if (q >= -(i32)reach && (i64)q < (i64)avail) w[b >> 2] |= (u32)src[q] << (8 * (b & 3));
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    };
    if (t < 2) S.cand[DEEP_THREADS + t] = 0;
    if (t == 0) { S.rep = 0; S.probe_good = 0; }
    // ---- what is a match worth?  (zl_match_auto_kernel, zl_encode.hip, says it at length: a sequence with a new offset costs its
    // offset's and its literal run's extra bits, the literals it replaces cost what the chunk's bytes cost; where bytes are cheap --
    // i.i.d. data over a dozen symbols -- short new-offset matches lose bits, and the deep search finds MANY of them: behind a full
    // history every position of such data has its five-byte twin, and nothing longer.)  Order-0 entropy of 16 KiB sampled from the
    // chunk's four quarters; a cheap chunk (a part of one) looks at the sequences of its first window(s): where matches covered less
    // than a tenth of the bytes, or fewer than half of the sequences were REGULAR -- eight bytes long, nearer than 64 bytes, or at the offset
    // of the sequence before (a vector of doubles: seven-byte matches, every one at offset 8, one literal between them: ratio 8; cheap
    // bytes by their histogram) -- only new-offset matches of smin_cheap bytes and more are kept from there on.
    bool cheap = false;
    u32 smin_cheap = 8;
    if (n >= 16384) {
        u32* const bins = S.t_end;  // [256]
        bins[t] = 0;
        lds_barrier();
        for (u32 part = 0; part < 4; part++) {
            const uint4 x = ld128(src + (size_t)part * (n >> 2) + 16 * t);
            const u32 w4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; k++) { atomicAdd(&bins[w4[k] & 0xFF], 1u); atomicAdd(&bins[(w4[k] >> 8) & 0xFF], 1u); atomicAdd(&bins[(w4[k] >> 16) & 0xFF], 1u); atomicAdd(&bins[w4[k] >> 24], 1u); }
        }
        lds_barrier();
        const u32 cnt = bins[t];
        float bits = cnt ? (float)cnt * (14.0f - __log2f((float)cnt)) : 0.0f;  // cnt * log2(16384 / cnt)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) bits += __shfl_xor(bits, d, 64);
        if ((t & 63) == 0) S.scan_a[t >> 6] = __float_as_uint(bits);
        lds_barrier();
        const float total = __uint_as_float(S.scan_a[0]) + __uint_as_float(S.scan_a[1]) + __uint_as_float(S.scan_a[2]) + __uint_as_float(S.scan_a[3]);
        cheap = total * 10.0f < 36.0f * 16384.0f;  // below 3.6 bits per byte
        // ... and the shortest match that pays: a sequence with a new offset inside such data costs some 45 bits (the offset's and the
        // literal run's extra bits), the bytes it replaces cost total / 16384 bits each -- 14 bytes on ten equally likely symbols.  (The
        // table parse asks for 8 and finds two such matches per chunk; this search finds all eighty of them: 0.8 % of the output)
        if (cheap) { const u32 need = (u32)(45.0f * 16384.0f / fmaxf(total, 16384.0f)) + 1u; smin_cheap = need < 8u ? 8u : (need > 24u ? 24u : need); }
        lds_barrier();
    }
    u32 smin = MIN_MATCH, probed = 0, probed_lits = 0, probed_seqs = 0;  // shortest new-offset match kept (uniform)
    const u32 probe_len = parts > 1 ? 256u : 1024u;    // (every part of a shared chunk probes for itself: a shorter probe, or a quarter of the chunk keeps the short matches)
    u32 nseq = 0, nlit = 0, carry = 0;
    uint4 wb_cur = make_uint4(0, 0, 0, 0);
    if (t < DEEP_WBUF / 16) wb_cur = ld16((i32)(p_lo + 16 * t));
    u64 bw_cur = p_lo + t < p_hi ? best[p_lo + t] : 0ull;
    lds_barrier();
    for (u32 ip = p_lo; ip < p_hi; ip += DEEP_THREADS) {
        const u32 p = ip + t;
        const bool in_range = p < p_hi;
        const u32 rep = S.rep;
        if (t < DEEP_WBUF / 16) *(uint4*)(S.wbuf + 16 * t) = wb_cur;
        // the bytes the repeat offset points at, for the whole window: [ip - rep - 8, + DEEP_RBUF)
        if (rep && t < DEEP_RBUF / 16) *(uint4*)(S.rbuf + 16 * t) = ld16((i32)ip - (i32)rep - 8 + (i32)(16 * t));
        // the next window's bytes and words travel while this one is parsed
        uint4 wb_next = make_uint4(0, 0, 0, 0);
        u64 bw_next = 0;
        if (ip + DEEP_THREADS < p_hi) {
            if (t < DEEP_WBUF / 16) wb_next = ld16((i32)(ip + DEEP_THREADS + 16 * t));
            if (p + DEEP_THREADS < p_hi) bw_next = best[p + DEEP_THREADS];
        }
        lds_barrier();
        const u64 v = lds64(S.wbuf, t);
        // (the eight bytes in front of a position are what the lane eight positions down holds)
        const u64 v_front = (u64)__shfl_up((u32)v, 8, 64) | ((u64)__shfl_up((u32)(v >> 32), 8, 64) << 32);
        u32 best_len = 0, best_off = 0, back = 0;
        if (in_range && carry < DEEP_THREADS) {  // (otherwise the whole window lies inside an earlier match)
            const u32 room = p_hi - p;
            u32 len_d = (u32)bw_cur & 0xFFFFu;
            const u32 off_d = (u32)(bw_cur >> 32), back_d = ((u32)bw_cur >> 16) & 0xFFu;
            if (len_d > room) len_d = room;
            if (len_d < smin && (smin == MIN_MATCH || off_d >= 64u)) len_d = 0;  // (near offsets are cheap to name: kept at any length)
            // the repeat offset: measured out of the two LDS buffers
            u32 l_r = 0;
            if (rep && p + reach >= rep) {
                l_r = common_bytes(v, lds64(S.rbuf, t + 8));
                for (u32 o = 8; l_r == o && o < ZL_DEEP_CAP; o += 16) l_r += lds_common16(S.wbuf, t + o, S.rbuf, t + 8 + o);
                if (l_r > room) l_r = room;
            }
            // a repeat offset costs no offset bits and leaves the history as it is: it wins unless the other is clearly better
            const int g_d = len_d >= MIN_MATCH ? 4 * (int)len_d - (int)(31 - __clz(off_d + 1)) : -1000;
            const int g_r = l_r >= MIN_MATCH ? 4 * (int)l_r + 4 : -2000;
            if (g_r >= g_d || (off_d == rep && l_r >= MIN_MATCH)) {
                if (l_r >= MIN_MATCH) {
                    best_len = (off_d == rep && len_d > l_r) ? len_d : l_r;  // (the same match, measured further by the search)
                    best_off = rep;
                    if ((t & 63) >= 8) {
                        const u64 x = v_front ^ lds64(S.rbuf, t);
                        back = x ? (u32)(__builtin_clzll(x) >> 3) : 8u;
                        const u32 src_room = p + reach - rep;  // bytes between the section's (frame's) start and the source
                        if (back > src_room) back = src_room;
                    }
                }
            } else { best_len = len_d; best_off = off_d; back = back_d; }
        }
        bool has_match = best_len >= MIN_MATCH;
        S.cand[t] = has_match ? ((u64)best_len | ((u64)best_off << 32)) : 0ull;
        lds_barrier();
        // ---- lazy evaluation against the matches one and two positions on (zstd.c:31546 ZSTD_compressBlock_lazy_generic: a gain of 4 / 7 is asked for)
        if (has_match) {
            const int g0 = 4 * (int)best_len - (int)(31 - __clz(best_off + 1));
            const u64 n1 = S.cand[t + 1];
            if (n1 && 4 * (int)(u32)n1 - (int)(31 - __clz((u32)(n1 >> 32) + 1)) > g0 + 4) has_match = false;
            const u64 n2 = S.cand[t + 2];
            if (a.lazy2 && n2 && 4 * (int)(u32)n2 - (int)(31 - __clz((u32)(n2 >> 32) + 1)) > g0 + 7) has_match = false;
        }
        // ---- greedy parse over the window: wave-local closure with lane shuffles, the four segments chained through LDS
        const u32 wbase = t & ~63u, wave = t >> 6;
        u32 jr[6];
        u32 jp = t + (has_match ? best_len : 1u);
#pragma unroll
        for (int r = 0; r < 6; r++) {
            jr[r] = jp;
            const u32 far = __shfl(jp, (int)((jp - wbase) & 63), 64);
            jp = (jp < wbase + 64) ? far : jp;
        }
        S.jump[t] = (u16)jp;
        lds_barrier();
        u32 enter = carry, my_entry = 0xFFFFFFFFu;
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            if (enter < 64 * (k + 1)) {
                if (k == wave) my_entry = enter;
                enter = S.jump[enter];
            }
        }
        carry = enter - DEEP_THREADS;
        const bool on_path = zl_path_marks(jr, t, wbase, my_entry);
        bool is_seq = on_path && has_match, is_lit = on_path && !has_match && in_range;
        // ---- backward extension over the free literals in front of a sequence, inside its wave (zstd.c:31707 "catch up")
        u32 bk = 0;
        {
            const u64 ms0 = __ballot(is_seq), ml0 = __ballot(is_lit);
            const u32 lane = t & 63;
            if (is_seq && back) {
                const u64 below_inv = ~ml0 & ((1ull << lane) - 1);                                  // lanes below that are NOT literals
                const u32 run = below_inv ? lane - (64u - (u32)__builtin_clzll(below_inv)) : lane;  // free literals right in front of this lane
                bk = back < run ? back : run;
            }
            // a literal lane asks the next sequence lane above it how far back that one reaches
            const u64 above = lane == 63 ? 0ull : (ms0 >> (lane + 1));
            const u32 nxt_seq = above ? lane + 1 + (u32)__builtin_ctzll(above) : lane;
            const u32 k_nxt = __shfl(bk, (int)nxt_seq, 64);
            if (is_lit && above && k_nxt >= nxt_seq - lane) is_lit = false;
        }
        // ---- compaction ranks from ballots; the four waves exchange their totals through LDS
        const u64 ms = __ballot(is_seq), ml = __ballot(is_lit);
        const u64 below = (1ull << (t & 63)) - 1;
        if (cheap && probed < probe_len) {  // (uniform) the probe of the cheap-bytes rule: which of this window's sequences are regular?
            const u32 lane = t & 63;
            const u64 lower = ms & below;
            const u32 prev = lower ? 63u - (u32)__builtin_clzll(lower) : lane;
            const u32 off_prev = __shfl(best_off, (int)prev, 64);
            const u64 mg = __ballot(is_seq && (best_len >= 8 || best_off < 64u || best_off == rep || (lower && off_prev == best_off)));
            if (lane == 0) atomicAdd(&S.probe_good, (u32)__popcll(mg));
        }
        if ((t & 63) == 0) S.scan_a[wave] = (u32)__popcll(ms) | ((u32)__popcll(ml) << 16);
        lds_barrier();
        u32 pre = 0, tot = 0;
#pragma unroll
        for (u32 w = 0; w < 4; w++) {
            const u32 x = S.scan_a[w];
            if (w < wave) pre += x;
            tot += x;
        }
        const u32 ex_s = (pre & 0xFFFF) + (u32)__popcll(ms & below), ex_l = (pre >> 16) + (u32)__popcll(ml & below);
        const u32 tot_s = tot & 0xFFFF, tot_l = tot >> 16;
        if (is_lit) lits[nlit + ex_l] = (u8)v;
        if (is_seq) {
            zl_seq sq;
            sq.lit_cum = nlit + ex_l;
            sq.match_len = best_len + bk;
            sq.offset = best_off;
            sq.pos = p - bk;
            seqs[nseq + ex_s] = sq;
            if (ex_s + 1 == tot_s) S.rep = best_off;  // last sequence of the window
        }
        nseq += tot_s;
        nlit += tot_l;
        if (cheap && probed < probe_len) {
            probed += DEEP_THREADS; probed_lits += tot_l; probed_seqs += tot_s;
            if (probed >= probe_len && ((probed - probed_lits) * 10 < probed || S.probe_good * 2 < probed_seqs)) smin = smin_cheap;
        }
        wb_cur = wb_next;
        bw_cur = bw_next;
        lds_barrier();
    }
    if (parts > 1) {
        if (t == 0) { a.sub_meta[2 * blockIdx.x] = nseq; a.sub_meta[2 * blockIdx.x + 1] = nlit; }
        return;
    }
    const u32 out_n = zl_merge_pieces(seqs, nseq, S.t_end, S.t_off, S.scan_a);
    if (t == 0) {
        a.meta[c].nseq = out_n;
        a.meta[c].nlit = nlit;
    }
}

// a.sub > 1: one workgroup per chunk puts the parts' literals and sequences behind one another where the entropy stage reads them
// (literal counts become the chunk's running count), joins a match cut at a part's end with its continuation, and leaves the counts
__global__ __launch_bounds__(DEEP_THREADS) void zl_deep_join_kernel(zl_match_args a) {
    __shared__ u32 t_end[DEEP_THREADS], t_off[DEEP_THREADS], scan_a[4];
    const u32 t = threadIdx.x, c = blockIdx.x, parts = a.sub;
    const u32* const sm = a.sub_meta + (size_t)2 * c * parts;
    u8* const lits = a.lits + (size_t)c * a.lit_stride;
    zl_seq* const seqs = a.seqs + (size_t)c * a.seq_stride;
    u32 nseq = 0, nlit = 0;
    for (u32 f = 0; f < parts; f++) nseq += sm[2 * f];
    u32 s_at = 0;
    for (u32 f = 0; f < parts; f++) {
        const u32 ns = sm[2 * f], nl = sm[2 * f + 1];
        const zl_seq* const from = a.sub_seqs + (size_t)c * a.seq_stride + (size_t)f * (a.seq_stride / parts);
        for (u32 i = t; i < ns; i += DEEP_THREADS) {
            zl_seq sq = from[i];
            sq.lit_cum += nlit;
            seqs[s_at + i] = sq;
        }
        if (nseq) {  // (a chunk without a sequence: the entropy stage reads its literals from the source)
            const u8* const lf = a.sub_lits + (size_t)c * a.lit_stride + f * (ZL_CHUNK / parts);
            for (u32 i = 16 * t; i < nl; i += 16 * DEEP_THREADS) {
                if (i + 16 <= nl) { const uint4 x = ld128(lf + i); __builtin_memcpy(lits + nlit + i, &x, 16); }
                else for (u32 b = i; b < nl; b++) lits[nlit + b] = lf[b];
            }
        }
        s_at += ns;
        nlit += nl;
    }
    const u32 out_n = zl_merge_pieces(seqs, nseq, t_end, t_off, scan_a);
    if (t == 0) {
        a.meta[c].nseq = out_n;
        a.meta[c].nlit = nlit;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
namespace {
// hipcub's temporary storage for sorting up to n keys: asked for once per power of two (the query walks rocPRIM's configuration
// machinery: not something a call of a few hundred microseconds wants to repeat) and kept
// (the figure of 2^lg items is taken for every n up to 2^lg -- rocPRIM picks its configuration by size and promises no
// monotonicity.  Should a sort ever refuse its storage, zl_launch_deep_search asks for that n exactly and raises `sort_floor`, which every
// later answer includes; the caller then sizes the scratch again.  0: the query itself failed -- no deep search.)
std::atomic<size_t> sort_floor{0};
size_t sort_query(u32 n) {
    size_t b = 0;
    if (hipcub::DeviceRadixSort::SortKeys(nullptr, b, (const u64*)nullptr, (u64*)nullptr, n, 32, 32 + ZL_DEEP_HLOG, (hipStream_t) nullptr) != hipSuccess) return 0;
    return (std::max<size_t>(b, 1) + 255) & ~(size_t)255;
}
size_t sort_temp_bytes(u32 n) {
    static std::atomic<size_t> known[33];
    u32 lg = 12;
    while (lg < 32 && ((u64)1 << lg) < n) lg++;
    size_t b = known[lg].load(std::memory_order_acquire);
    if (!b) {
        b = sort_query((u32)std::min<u64>((u64)1 << lg, DEEP_SEC_MAX));
        if (!b) return 0;
        known[lg].store(b, std::memory_order_release);
    }
    return std::max(b, sort_floor.load(std::memory_order_acquire));
}
// the longest section a search over `span` bytes sorts (sections end on multiples of DEEP_SEC_BYTES of the input; the first may be any part of one)
inline u32 sec_cap(u64 span) { return (u32)std::min<u64>(std::max<u64>(span, 4096), DEEP_SEC_MAX); }
}  // namespace

// two key arrays of the longest section + the sort's temporary storage (with room to spare: a shorter last section may ask for a little more)
extern "C" size_t zl_deep_scratch_bytes(u64 span) {
    const u32 cap = sec_cap(span);
    const size_t temp = sort_temp_bytes(cap);
    if (!temp) return 0;
    return 2 * (size_t)cap * sizeof(u64) + temp + (64u << 10) + 256;
}

extern "C" hipError_t zl_launch_deep_search(const zl_deep_args* a, hipStream_t stream) {
    if (a->hi <= a->lo) return hipSuccess;
    if (a->scratch_bytes == 0 || a->scratch_bytes < zl_deep_scratch_bytes(a->hi - a->lo)) return hipErrorInvalidValue;
    const u32 cap = sec_cap(a->hi - a->lo);
    u64* const keys_a = (u64*)a->scratch;
    u64* const keys_b = keys_a + cap;
    void* const temp = keys_b + cap;
    const size_t temp_room = a->scratch_bytes - 2 * (size_t)cap * sizeof(u64);
    const u32 mml_shift = 64u - 8u * a->mml;
    for (u64 s0 = a->lo; s0 < a->hi;) {
        const u64 s1 = std::min<u64>(a->hi, (s0 & ~(DEEP_SEC_BYTES - 1)) + DEEP_SEC_BYTES);
        const u32 n = (u32)(s1 - s0);
        const u8* sec = a->src + s0;
        const u32 blocks = (n + DEEP_THREADS - 1) / DEEP_THREADS;
        hipLaunchKernelGGL(zl_deep_key_kernel, dim3(blocks), dim3(DEEP_THREADS), 0, stream, sec, n, mml_shift, keys_a);
        size_t tb = temp_room;
        const hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, tb, (const u64*)keys_a, keys_b, n, 32, 32 + ZL_DEEP_HLOG, stream);
        if (e != hipSuccess) {
            // too little storage for THIS n (see sort_temp_bytes)?  Then say what it takes and let the caller come back with it
            const size_t need = sort_query(n);
            if (need > temp_room) {
                size_t f = sort_floor.load();
                while (f < need && !sort_floor.compare_exchange_weak(f, need)) {}
                (void)hipGetLastError();
                return hipErrorOutOfMemory;
            }
            return e;
        }
        u64* const best = a->best + (s0 - a->lo);
        u64* const chain = a->chain ? a->chain + (size_t)ZL_DEEP_CHAIN * (s0 - a->lo) : nullptr;
        if (a->k <= 16) hipLaunchKernelGGL((zl_deep_search_kernel<16>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, a->k, a->window, a->mml, a->frames, best, chain);
        else if (a->k <= 64) hipLaunchKernelGGL((zl_deep_search_kernel<64>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, a->k, a->window, a->mml, a->frames, best, chain);
        else if (a->k <= 128) hipLaunchKernelGGL((zl_deep_search_kernel<128>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, a->k, a->window, a->mml, a->frames, best, chain);
        else hipLaunchKernelGGL((zl_deep_search_kernel<256>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, std::min<u32>(a->k, 256u), a->window, a->mml, a->frames, best, chain);
        hipLaunchKernelGGL(zl_deep_extend_kernel, dim3(blocks), dim3(DEEP_THREADS), 0, stream, best, n, sec, chain);
        if (a->k2) {
            // levels 16+: the section once more, sorted by a hash of mml2 bytes -- the K most recent twins of a LONG prefix reach where the K most recent
            // of minMatch bytes end (a word of a small vocabulary has thousands of twins, the phrase around it a few) --, merged into the chains
            hipLaunchKernelGGL(zl_deep_key_kernel, dim3(blocks), dim3(DEEP_THREADS), 0, stream, sec, n, 64u - 8u * a->mml2, keys_a);
            size_t tb2 = temp_room;
            const hipError_t e2 = hipcub::DeviceRadixSort::SortKeys(temp, tb2, (const u64*)keys_a, keys_b, n, 32, 32 + ZL_DEEP_HLOG, stream);
            if (e2 != hipSuccess) return e2;
            const u32 fl = a->frames | 2u;
            if (a->k2 <= 64) hipLaunchKernelGGL((zl_deep_search_kernel<64>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, a->k2, a->window, a->mml2, fl, best, chain);
            else if (a->k2 <= 128) hipLaunchKernelGGL((zl_deep_search_kernel<128>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, a->k2, a->window, a->mml2, fl, best, chain);
            else hipLaunchKernelGGL((zl_deep_search_kernel<256>), dim3(blocks), dim3(DEEP_THREADS), 0, stream, (const u64*)keys_b, n, sec, std::min<u32>(a->k2, 256u), a->window, a->mml2, fl, best, chain);
            hipLaunchKernelGGL(zl_deep_extend_kernel, dim3(blocks), dim3(DEEP_THREADS), 0, stream, best, n, sec, chain);  // (a best word the second pass wrote may fill its record too)
        }
        s0 = s1;
    }
    return hipGetLastError();
}

extern "C" hipError_t zl_launch_match_deep(const zl_match_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    const u32 parts = a->sub > 1 ? a->sub : 1u;
    hipLaunchKernelGGL(zl_match_deep_kernel, dim3(nchunks * parts), dim3(DEEP_THREADS), 0, stream, *a, nchunks);
    if (parts > 1) hipLaunchKernelGGL(zl_deep_join_kernel, dim3(nchunks), dim3(DEEP_THREADS), 0, stream, *a);
    return hipGetLastError();
}
