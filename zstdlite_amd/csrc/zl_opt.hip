// zl_opt.hip -- the price-based parse behind levels 16 ... 22 for gfx950 (MI355X).
//
// The reference's levels 16-22 are its optimal parsers (zstd.c:34586-36163 ZSTD_compressBlock_opt_generic: btopt, btultra,
// btultra2): at every position a binary tree hands over ALL matches worth having -- one per length, the nearest offset that
// reaches it --, a forward pass over a table of arrival prices keeps for every position the cheapest way to get there (a literal
// more, or a match of any length from any position in front, the three repeat offsets tried first), and the chosen path is read
// off backwards.  A price is what a choice's codes cost under the symbol statistics gathered so far; btultra2 parses a block
// twice to have statistics worth the name.
//
// Here: the deep match finder (zl_deep.hip) already knows every position's K most recent twins.  For these levels its search
// kernel also leaves the CHAIN of a position -- among those K, nearest first, every one that is longer than all nearer ones: up
// to OPT_TOPC (length, offset) pairs, exactly "the nearest offset per length" --, the lazy parse of level 15 runs once to have
// sequences to count, and then one WAVE per chunk does what the reference does per block:
//
//   statistics   histograms of the chunk's literal bytes and of its sequences' literal-length, match-length and offset codes
//                (from the parse in front: level 15's first, then this kernel's own) -> prices in 1/32 bit: -log2(share) + the
//                code's extra bits (zstd.c:34700 ZSTD_setBasePrices / :34810 ZSTD_getMatchPrice).  A chunk owns its statistics:
//                the model (scripts/dev/sim/opt2_sim.c) is BETTER with them per 128 KiB than over the whole input.
//   forward      position by position.  A node = {price, literals pending, the three repeat offsets}; the node in hand pays a
//                literal into the next position and pushes every (candidate, length) -- one LANE per length, the chain entry that
//                covers it -- into a ring of the next OPT_RING positions' best match arrivals (a 64-bit LDS minimum: price above
//                length above offset); the three repeat offsets are measured against the bytes under the position out of four
//                256-byte windows the wave keeps in registers (the input under the position and under each repeat offset: they
//                advance together, so they are loaded once per 192 positions and when the history changes) and pushed likewise.
//                A match longer than OPT_CAP is taken whole (zstd.c:35020: sufficient_len).  Every final node leaves its arrival
//                {literal run | length, offset} in memory.
//   backward     from the chunk's end through those arrivals: the sequences, last first.
//   emit         the sequences turned round, literal counts summed (wave scans), literals copied; zl_opt_merge_kernel joins what
//                a match longer than its measure arrived as (zl_merge_pieces).
//
// CPU model with the reference's entropy stage behind it: scripts/dev/sim/opt2_sim.c (python sources, compressed size relative to
// the reference AT THAT LEVEL: 1.002 at 16, 1.013 at 17, 1.033 at 19, 1.026 at 22; round 5's level-15 parse at these levels: 1.07 - 1.11).
#include <hip/hip_runtime.h>

#include "zl_device.h"
#include "zl_encode.h"
#include "zl_format.h"
#include "zl_kernels.h"
#include "zl_match_common.h"

#define OPT_THREADS 64
#define OPT_TOPC ZL_DEEP_CHAIN
#define OPT_RING 256u      // positions ahead of the node in hand whose best match arrival is kept
#define OPT_CAP 192u       // lengths pushed one by one; a longer candidate is taken whole
#define OPT_MINM 4u        // shortest match (zstd.c:34990: 3 only where the level's minMatch is 3; the model prefers 4 on every input tried)
#define OPT_FRAC 5         // fraction bits of a price
#define OPT_NONE (~0ull)
#define OPT_SEC_BYTES ((u64)ZL_CHUNK << ZL_FAR_SECTION_LOG)

namespace {

__device__ __forceinline__ u32 opt_hb(u32 v) { return 31u - (u32)__clz(v); }
// code numbers (zstd.c:15470 ZSTD_LLcode, :15487 ZSTD_MLcode), as arithmetic
__device__ __forceinline__ u32 opt_ll_code(u32 l) {
    if (l < 16) return l;
    if (l < 24) return 16 + ((l - 16) >> 1);
    if (l < 32) return 20 + ((l - 24) >> 2);
    if (l < 48) return 22 + ((l - 32) >> 3);
    if (l < 64) return 24;
    return opt_hb(l) + 19;
}
__device__ __forceinline__ u32 opt_ml_code(u32 mlb) {  // mlb = match length - 3
    if (mlb < 32) return mlb;
    if (mlb < 40) return 32 + ((mlb - 32) >> 1);
    if (mlb < 48) return 36 + ((mlb - 40) >> 2);
    if (mlb < 64) return 38 + ((mlb - 48) >> 3);
    if (mlb < 96) return 40 + ((mlb - 64) >> 4);
    if (mlb < 128) return 42;
    return opt_hb(mlb) + 36;
}
__device__ __forceinline__ u32 opt_ll_bits(u32 c) { return c < 16 ? 0u : c < 20 ? 1u : c < 22 ? 2u : c < 24 ? 3u : c == 24 ? 4u : c + 6u - 25u; }   // 25 -> 6, ... 35 -> 16
__device__ __forceinline__ u32 opt_ml_bits(u32 c) { return c < 32 ? 0u : c < 36 ? 1u : c < 38 ? 2u : c < 40 ? 3u : c < 42 ? 4u : c == 42 ? 5u : c + 7u - 43u; }  // 43 -> 7, ... 52 -> 16

struct OptShared {
    union {
        u64 arr[OPT_RING];   // forward pass: best match arrival per position of the ring: price << 36 | length << 28 | offset
        u32 hist[384];       // statistics: 256 literal bytes, 36 literal-length codes at 256, 53 match-length codes at 292, 32 offset codes at 345
    };
    u32 rep[OPT_RING][3];    // the repeat offsets of every final node; [0] bit 31: no literal pending at the node
    u16 p_lit[256], p_ll[36], p_ml[53], p_of[32];
    u32 tot[4];
};

struct OptRec { u32 pos, len, off, litlen; };  // a sequence as the backward pass finds it (stored in the chunk's own sequence array, from its end down)

// 4 bytes at window offset o (o + 4 <= 256) of a window whose lane j holds the dword at byte 4 j; o is uniform
__device__ __forceinline__ u32 win4(u32 win, u32 o) {
    const u32 k = (u32)__builtin_amdgcn_readfirstlane((int)(o >> 2));
    const u32 d0 = (u32)__builtin_amdgcn_readlane((int)win, (int)k), d1 = (u32)__builtin_amdgcn_readlane((int)win, (int)((k + 1) & 63));
    return __builtin_amdgcn_alignbyte(d1, d0, o & 3);
}
// the byte at window offset o + lane
__device__ __forceinline__ u32 win_byte(u32 win, u32 o, u32 lane) {
    const u32 at = o + lane;
    const u32 d = (u32)__shfl((int)win, (int)((at >> 2) & 63), 64);
    return (d >> (8 * (at & 3))) & 0xFFu;
}

}  // namespace

// chain: OPT_TOPC words per input byte from position deep_lo on; word: [15:0] length, [63:32] offset, 0: none; ascending lengths, the longest in slot OPT_TOPC - 1
// arrivals: one word per input byte (the buffer the lazy parse read its candidates from: it is done with them)
__global__ __launch_bounds__(OPT_THREADS) void zl_opt_parse_kernel(zl_match_args a, const u64* __restrict__ chain_all, u64* __restrict__ arrivals_all, u32 nchunks, u32 passes) {
    __shared__ OptShared S;
    const u32 lane = threadIdx.x;
    const u32 c = a.order ? a.order[zl_spread(blockIdx.x, nchunks)] : zl_spread(blockIdx.x, nchunks);
    const zl_chunk_desc cd = a.chunks[c];
    const u8* const src = a.src + cd.src_off;
    const u32 n = cd.src_size;
    const u64* const chain = chain_all + (size_t)(cd.src_off - a.deep_lo) * OPT_TOPC;
    u64* const arrivals = arrivals_all + (size_t)(cd.src_off - a.deep_lo);
    if (n < 16) return;  // (nothing to gain; the lazy parse's sequences stand)
    const u64 sec_off = cd.src_off & (OPT_SEC_BYTES - 1);
    const u32 reach = a.joined ? (u32)sec_off : 0u;  // bytes in front of the chunk a source may lie in
    const u64 sec_end = cd.src_off - sec_off + OPT_SEC_BYTES;
    const u32 avail = (u32)((a.total < sec_end ? a.total : sec_end) - cd.src_off);  // bytes readable from the chunk's first byte on
    // a window: lane j holds the dword at chunk offset base + 4 j (base may be negative: in front of the chunk); outside [-reach, avail): zero
    auto load_win = [&](i32 base) -> u32 {
        const i32 at = base + 4 * (i32)lane;
        u32 v = 0;
        if (at >= -(i32)reach && (i64)at + 4 <= (i64)avail) __builtin_memcpy(&v, src + at, 4);
        else for (u32 b = 0; b < 4; b++) { const i32 q = at + (i32)b; if (q >= -(i32)reach && (i64)q < (i64)avail) v |= (u32)src[q] << (8 * b); }
        return v;
    };
    // LDS traffic of ONE wave needs no fence: the LDS serves a wave's operations in the order it issued them; what must not happen is the
    // compiler moving them (a workgroup-scope release fence would also wait for every global store in flight: once per position, a microsecond each)
    auto lds_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    // The two places a parse of this chunk can lie: the launch's own arrays (where the lazy parse left its sequences and where the entropy
    // stage will look) and the arrays a shared chunk's parts use at the lower levels (free here).  A pass writes into the one that does
    // not hold the best parse so far.
    u8* const lit_buf[2] = {a.lits + (size_t)c * a.lit_stride, a.sub_lits + (size_t)c * a.lit_stride};
    zl_seq* const seq_buf[2] = {a.seqs + (size_t)c * a.seq_stride, a.sub_seqs + (size_t)c * a.seq_stride};
    // statistics of a parse into S.hist; returns what the parse costs under ITS OWN statistics, in bits: every count times the
    // information of its share, plus the codes' extra bits -- what the entropy stage will make of it, near enough to compare two parses
    auto take_stats = [&](const u8* L, const zl_seq* Q, u32 ns, u32 nl) -> float {
        for (u32 i = lane; i < 384; i += OPT_THREADS) S.hist[i] = 0;
        lds_sync();
        const u8* const lsrc = ns ? L : src;  // (a chunk without a sequence keeps its literals in the source)
        for (u32 i = 16 * lane; i < nl; i += 16 * OPT_THREADS) {
            if (i + 16 <= nl) {
                const uint4 x = ld128(lsrc + i);
                const u32 w4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int k = 0; k < 4; k++) { atomicAdd(&S.hist[w4[k] & 0xFF], 1u); atomicAdd(&S.hist[(w4[k] >> 8) & 0xFF], 1u); atomicAdd(&S.hist[(w4[k] >> 16) & 0xFF], 1u); atomicAdd(&S.hist[w4[k] >> 24], 1u); }
            } else for (u32 b = i; b < nl; b++) atomicAdd(&S.hist[lsrc[b]], 1u);
        }
        for (u32 i = lane; i < ns; i += OPT_THREADS) {
            const zl_seq sq = Q[i];
            const u32 prev_cum = i ? Q[i - 1].lit_cum : 0u;
            const u32 o1 = i ? Q[i - 1].offset : 0u, o2 = i > 1 ? Q[i - 2].offset : 0u, o3 = i > 2 ? Q[i - 3].offset : 0u;
            // (the history as the three offsets in front: near enough for a count)
            const u32 oc = (i && sq.offset == o1 && sq.lit_cum != prev_cum) ? 0u : ((i > 1 && sq.offset == o2) || (i > 2 && sq.offset == o3)) ? 1u : opt_hb(sq.offset + 3);
            atomicAdd(&S.hist[256 + opt_ll_code(sq.lit_cum - prev_cum)], 1u);
            atomicAdd(&S.hist[292 + opt_ml_code(sq.match_len - 3)], 1u);
            atomicAdd(&S.hist[345 + (oc & 31)], 1u);
        }
        lds_sync();
        float t_lit = 0, t_ll = 0, t_ml = 0, t_of = 0, x_lit = 0, x_ll = 0, x_ml = 0, x_of = 0, extra = 0;  // totals; sums of h log2 h; extra bits
        for (u32 i = lane; i < 256; i += OPT_THREADS) { const float h = (float)S.hist[i]; t_lit += h; x_lit += h > 0 ? h * __log2f(h) : 0.0f; }
        if (lane < 36) { const float h = (float)S.hist[256 + lane]; t_ll = h; x_ll = h > 0 ? h * __log2f(h) : 0.0f; extra += h * (float)opt_ll_bits(lane); }
        if (lane < 53) { const float h = (float)S.hist[292 + lane]; t_ml = h; x_ml = h > 0 ? h * __log2f(h) : 0.0f; extra += h * (float)opt_ml_bits(lane); }
        if (lane < 32) { const float h = (float)S.hist[345 + lane]; t_of = h; x_of = h > 0 ? h * __log2f(h) : 0.0f; extra += h * (float)lane; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            t_lit += __shfl_xor(t_lit, d, 64); t_ll += __shfl_xor(t_ll, d, 64); t_ml += __shfl_xor(t_ml, d, 64); t_of += __shfl_xor(t_of, d, 64);
            x_lit += __shfl_xor(x_lit, d, 64); x_ll += __shfl_xor(x_ll, d, 64); x_ml += __shfl_xor(x_ml, d, 64); x_of += __shfl_xor(x_of, d, 64); extra += __shfl_xor(extra, d, 64);
        }
        auto ent = [](float t, float x) { return t > 0 ? t * __log2f(t) - x : 0.0f; };
        return ent(t_lit, x_lit) + ent(t_ll, x_ll) + ent(t_ml, x_ml) + ent(t_of, x_of) + extra;
    };
    u32 best_buf = 0, best_nseq = a.meta[c].nseq, best_nlit = a.meta[c].nlit;
    float best_cost = take_stats(lit_buf[0], seq_buf[0], best_nseq, best_nlit);
    for (u32 pass = 0; pass < passes; pass++) {
        u8* const lits = lit_buf[best_buf ^ 1];
        zl_seq* const seqs = seq_buf[best_buf ^ 1];
        u32 nseq = 0, nlit = 0;
        // ------------------------------------------------------------------ prices from the statistics of the best parse so far
        {
            // totals (every bin counts a quarter more than it saw: nothing is impossible)
            float s_lit = 0, s_ll = 0, s_ml = 0, s_of = 0;
            for (u32 i = lane; i < 256; i += OPT_THREADS) s_lit += (float)S.hist[i] + 0.25f;
            if (lane < 36) s_ll = (float)S.hist[256 + lane] + 0.25f;
            if (lane < 53) s_ml = (float)S.hist[292 + lane] + 0.25f;
            if (lane < 32) s_of = (float)S.hist[345 + lane] + 0.25f;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { s_lit += __shfl_xor(s_lit, d, 64); s_ll += __shfl_xor(s_ll, d, 64); s_ml += __shfl_xor(s_ml, d, 64); s_of += __shfl_xor(s_of, d, 64); }
            const float unit = (float)(1 << OPT_FRAC);
            auto price = [&](u32 h, float total, u32 extra) -> u16 {
                const float bits = __log2f(total) - __log2f((float)h + 0.25f) + (float)extra;
                const float q = bits * unit + 0.5f;
                return (u16)(q < 1.0f ? 1.0f : (q > 60000.0f ? 60000.0f : q));
            };
            u16 pl[4], pll = 0, pml = 0, pof = 0;
            for (u32 k = 0; k < 4; k++) pl[k] = price(S.hist[lane + 64 * k], s_lit, 0);
            if (lane < 36) pll = price(S.hist[256 + lane], s_ll, opt_ll_bits(lane));
            if (lane < 53) pml = price(S.hist[292 + lane], s_ml, opt_ml_bits(lane));
            if (lane < 32) pof = price(S.hist[345 + lane], s_of, lane);
            lds_sync();
            for (u32 k = 0; k < 4; k++) S.p_lit[lane + 64 * k] = pl[k];
            if (lane < 36) S.p_ll[lane] = pll;
            if (lane < 53) S.p_ml[lane] = pml;
            if (lane < 32) S.p_of[lane] = pof;
        }
        // (hist shares its LDS with the ring: prices are in registers / their own arrays before the ring is filled)
        for (u32 i = lane; i < OPT_RING; i += OPT_THREADS) S.arr[i] = OPT_NONE;
        lds_sync();
        // ------------------------------------------------------------------ forward
        // Per lane, once per pass: the match-length price of the length this lane stands for in a first round of pushes, and the tables a
        // uniform look-up reads with v_readlane instead of an LDS round trip
        const u32 pml_lane = S.p_ml[opt_ml_code(OPT_MINM + lane - 3)];
        const u32 v_pll = lane < 36 ? (u32)S.p_ll[lane] : 0u;
        const u32 pll0 = S.p_ll[0], pof0 = S.p_of[0], pof1 = S.p_of[1];
        auto pll_of = [&](u32 litlen) -> u32 { const u32 cc = opt_ll_code(litlen); return (u32)__builtin_amdgcn_readlane((int)v_pll, (int)__builtin_amdgcn_readfirstlane((int)(cc > 35 ? 35 : cc))); };
        u32 i = 0;                       // the node in hand (chunk offset)
        u32 price = pll0;                // what it costs to stand here, the pending literal run's length code included
        u32 litlen = 0;                  // literals pending
        u32 pll_cur = pll0;              // ... and what that run's length code costs
        u32 r0 = 0, r1 = 0, r2 = 0;      // the history (0: none yet)
        u64 cur_w = 0x80000000ull;       // how the node in hand was reached (stored when something will ask: it arrived by a match, or it pushes)
        bool cur_stored = true;
        i32 wb = 0;                      // window base (chunk offset, a multiple of 4)
        u32 w_cur = load_win(0), w_r0 = 0, w_r1 = 0, w_r2 = 0;
        u32 cb = 0;                      // chain batch: lane l holds word l & 3 of position cb + (l >> 2)
        u64 cw = chain[lane];
        u64 cw_next = (size_t)64 + lane < (size_t)n * OPT_TOPC ? chain[64 + lane] : 0ull;
        auto word_price = [&](u64 w) -> u32 { const u32 off = (u32)(w >> 32); return off ? (u32)S.p_of[opt_hb(off + 3) & 31] : 0u; };
        u32 cpo = word_price(cw);        // the offset-code price of every word of the batch
        u64 candmask = __ballot(((u32)cw & 0xFFFFu) >= OPT_MINM);  // bit 4 k + 3: position cb + k has a candidate
        u64 end_w = 0x80000000ull;       // how the chunk's end is reached (position n's word would be the next chunk's first: kept here)
        u32 far = 0;                     // the farthest position a pushed match lands on
        // per 64 positions from mb on: where each repeat offset matches four bytes, and every position's literal price
        u32 mb = 0, plit64 = 0;
        u64 rm0 = 0, rm1 = 0, rm2 = 0;
        bool dirty = true;
        while (i < n) {
            // ---- the windows, the per-position masks and the chain batch keep up with the position
            if (dirty || i - mb >= 64u) {
                if (i - (u32)wb > 124u) {
                    wb = (i32)(i & ~3u);
                    w_cur = load_win(wb);
                    w_r0 = r0 ? load_win(wb - (i32)r0) : 0u; w_r1 = r1 ? load_win(wb - (i32)r1) : 0u; w_r2 = r2 ? load_win(wb - (i32)r2) : 0u;
                }
                mb = i;
                const u32 x = i - (u32)wb + lane;  // this lane's position inside the windows (x + 4 <= 256)
                const u32 k0 = x >> 2, sh = x & 3;
                auto lane4 = [&](u32 win) -> u32 { const u32 d0 = (u32)__shfl((int)win, (int)(k0 & 63), 64), d1 = (u32)__shfl((int)win, (int)((k0 + 1) & 63), 64); return __builtin_amdgcn_alignbyte(d1, d0, sh); };
                const u32 c4 = lane4(w_cur);
                const bool room4 = i + lane + OPT_MINM <= n;
                rm0 = r0 ? __ballot(room4 && lane4(w_r0) == c4) : 0ull;
                rm1 = r1 ? __ballot(room4 && lane4(w_r1) == c4) : 0ull;
                rm2 = r2 ? __ballot(room4 && lane4(w_r2) == c4) : 0ull;
                plit64 = S.p_lit[c4 & 0xFFu];
                dirty = false;
            }
            if (i - cb >= 16u) {
                if (i - cb < 32u) { cb += 16; cw = cw_next; }
                else { cb = i & ~15u; cw = (size_t)cb * OPT_TOPC + lane < (size_t)n * OPT_TOPC ? chain[(size_t)cb * OPT_TOPC + lane] : 0ull; }
                const size_t nx = (size_t)(cb + 16) * OPT_TOPC + lane;
                cw_next = nx < (size_t)n * OPT_TOPC ? chain[nx] : 0ull;
                cpo = word_price(cw);
                candmask = __ballot(((u32)cw & 0xFFFFu) >= OPT_MINM);
            }
            const u32 j = (u32)__builtin_amdgcn_readfirstlane((int)(i - mb));
            const u32 q = (u32)__builtin_amdgcn_readfirstlane((int)(4u * (i - cb)));
            const u32 room = n - i;
            const bool ll0 = litlen == 0;
            const u32 base = price;
            // which repeat offsets are worth measuring here (zstd.c:19680: with no literal pending the codes mean the second, the third, and the
            // first less one -- the last is not tried)
            const bool t0 = !ll0 && ((rm0 >> j) & 1), t1 = ((rm1 >> j) & 1) && !(ll0 && r1 == r0), t2 = ((rm2 >> j) & 1) && !(ll0 && r2 == r0);
            const bool has_c = ((candmask >> (q + 3)) & 1) && room >= OPT_MINM;
            u32 jump_len = 0, jump_off = 0;
            if (t0 || t1 || t2 || has_c) {
                // this node is somebody's source: its history and how it was reached are kept
                if (lane == 0) {
                    S.rep[i & (OPT_RING - 1)][0] = r0 | (ll0 ? 0x80000000u : 0u); S.rep[i & (OPT_RING - 1)][1] = r1; S.rep[i & (OPT_RING - 1)][2] = r2;
                    if (!cur_stored && i) arrivals[i] = cur_w;
                }
                cur_stored = true;
                const u32 o = i - (u32)wb;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    if (!(r == 0 ? t0 : r == 1 ? t1 : t2)) continue;  // (uniform)
                    const u32 off = r == 0 ? r0 : r == 1 ? r1 : r2;
                    const u32 wr = r == 0 ? w_r0 : r == 1 ? w_r1 : w_r2;
                    const bool differs = win_byte(w_cur, o, lane) != win_byte(wr, o, lane);
                    const u64 dm = __ballot(differs);
                    u32 L = dm ? (u32)__builtin_ctzll(dm) : 64u;
                    // (rare: the repeat runs on past the windows' 64 bytes -- measured on from memory, 64 bytes a round, so that it is priced as ONE match)
                    for (u32 more = 64; L == more && more < OPT_CAP && more < room; more += 64) {
                        const u32 at = i + more + lane;
                        const bool dif = at >= n || (u64)at >= (u64)avail || src[at] != src[(i32)at - (i32)off];
                        const u64 m2 = __ballot(dif);
                        L += m2 ? (u32)__builtin_ctzll(m2) : 64u;
                    }
                    if (L > room) L = room;
                    if (L > OPT_CAP) L = OPT_CAP;
                    if (L < OPT_MINM) continue;
                    const u32 code = ll0 ? (u32)r - 1u : (r == 0 ? 0u : 1u);  // (r = 1 with no literal pending is code 0, r = 2 code 1)
                    const u32 pc = base + (code ? pof1 : pof0) + pll0;
                    if (i + L > far) far = i + L;
                    {
                        const u32 l = OPT_MINM + lane;
                        if (l <= L) atomicMin((unsigned long long*)&S.arr[(i + l) & (OPT_RING - 1)], ((u64)(pc + pml_lane) << 36) | ((u64)l << 28) | off);
                    }
                    for (u32 lb = OPT_MINM + 64; lb <= L; lb += 64) {
                        const u32 l = lb + lane;
                        if (l <= L) atomicMin((unsigned long long*)&S.arr[(i + l) & (OPT_RING - 1)], ((u64)(pc + S.p_ml[opt_ml_code(l - 3)]) << 36) | ((u64)l << 28) | off);
                    }
                }
                // ---- the chain's candidates: lengths OPT_MINM .. the longest, each from the nearest entry that reaches it
                if (has_c) {
                    const u32 lo32 = (u32)cw, hi32 = (u32)(cw >> 32);
                    u32 L3 = (u32)__builtin_amdgcn_readlane((int)lo32, (int)(q + 3)) & 0xFFFFu;
                    const u32 O3 = (u32)__builtin_amdgcn_readlane((int)hi32, (int)(q + 3));
                    if (L3 > room) L3 = room;
                    if (L3 > OPT_CAP) { jump_len = L3; jump_off = O3; }
                    else {
                        const u32 L0 = (u32)__builtin_amdgcn_readlane((int)lo32, (int)q) & 0xFFFFu, L1 = (u32)__builtin_amdgcn_readlane((int)lo32, (int)(q + 1)) & 0xFFFFu, L2 = (u32)__builtin_amdgcn_readlane((int)lo32, (int)(q + 2)) & 0xFFFFu;
                        const u32 O0 = (u32)__builtin_amdgcn_readlane((int)hi32, (int)q), O1 = (u32)__builtin_amdgcn_readlane((int)hi32, (int)(q + 1)), O2 = (u32)__builtin_amdgcn_readlane((int)hi32, (int)(q + 2));
                        const u32 b0 = base + pll0;
                        const u32 pc0 = b0 + (u32)__builtin_amdgcn_readlane((int)cpo, (int)q), pc1 = b0 + (u32)__builtin_amdgcn_readlane((int)cpo, (int)(q + 1));
                        const u32 pc2 = b0 + (u32)__builtin_amdgcn_readlane((int)cpo, (int)(q + 2)), pc3 = b0 + (u32)__builtin_amdgcn_readlane((int)cpo, (int)(q + 3));
                        if (i + L3 > far) far = i + L3;
                        {
                            const u32 l = OPT_MINM + lane;
                            if (l <= L3) {
                                const u32 off = L0 >= l ? O0 : L1 >= l ? O1 : L2 >= l ? O2 : O3;
                                const u32 pc = L0 >= l ? pc0 : L1 >= l ? pc1 : L2 >= l ? pc2 : pc3;
                                atomicMin((unsigned long long*)&S.arr[(i + l) & (OPT_RING - 1)], ((u64)(pc + pml_lane) << 36) | ((u64)l << 28) | off);
                            }
                        }
                        for (u32 lb = OPT_MINM + 64; lb <= L3; lb += 64) {
                            const u32 l = lb + lane;
                            if (l <= L3) {
                                const u32 off = L0 >= l ? O0 : L1 >= l ? O1 : L2 >= l ? O2 : O3;
                                const u32 pc = L0 >= l ? pc0 : L1 >= l ? pc1 : L2 >= l ? pc2 : pc3;
                                atomicMin((unsigned long long*)&S.arr[(i + l) & (OPT_RING - 1)], ((u64)(pc + S.p_ml[opt_ml_code(l - 3)]) << 36) | ((u64)l << 28) | off);
                            }
                        }
                    }
                }
                lds_sync();
            }
            // ---- the next node: a match taken whole, or position i + 1 by a literal or by the best match that lands there
            u32 ni, nprice, nlitlen, nr0 = r0, nr1 = r1, nr2 = r2, m_len = 0, m_off = 0, m_src = i;
            bool src_ll0 = ll0;
            if (jump_len) {
                ni = i + jump_len;
                m_len = jump_len; m_off = jump_off;
                const u32 mc = opt_ml_code(jump_len - 3);
                nprice = base + pll0 + S.p_of[opt_hb(jump_off + 3) & 31] + S.p_ml[mc > 52 ? 52 : mc];
                nlitlen = 0;
                for (u32 k = lane; k < OPT_RING; k += OPT_THREADS) S.arr[k] = OPT_NONE;  // whatever was on its way into the match, or over it, is dropped (as the reference restarts its table behind such a match)
                far = ni;
            } else {
                ni = i + 1;
                const u32 pll_next = pll_of(litlen + 1);
                const i32 lit_price = (i32)base + (i32)__builtin_amdgcn_readlane((int)plit64, (int)j) + (i32)pll_next - (i32)pll_cur;
                u64 aw = OPT_NONE;
                if (ni <= far) {
                    aw = S.arr[ni & (OPT_RING - 1)];
                    if (aw != OPT_NONE && lane == 0) S.arr[ni & (OPT_RING - 1)] = OPT_NONE;
                }
                const u32 a_price = (u32)(aw >> 36);
                if (aw != OPT_NONE && (i32)a_price < lit_price) {
                    m_len = (u32)(aw >> 28) & 0xFFu; m_off = (u32)aw & 0x0FFFFFFFu;
                    m_src = ni - m_len;
                    nprice = a_price;
                    nlitlen = 0;
                    if (m_src != i) {
                        const u32 s0 = S.rep[m_src & (OPT_RING - 1)][0];
                        nr0 = s0 & 0x7FFFFFFFu; nr1 = S.rep[m_src & (OPT_RING - 1)][1]; nr2 = S.rep[m_src & (OPT_RING - 1)][2];
                        src_ll0 = (s0 >> 31) != 0;
                    }
                } else { nprice = (u32)lit_price; nlitlen = litlen + 1; pll_cur = pll_next; }
            }
            if (m_len) {
                // the history behind a match (by value: the decoder's rules, zstd.c:19680): the offset in hand stays, the second swaps, anything else pushes --
                // and with no literal in front the first cannot be named as a repeat: it is a new offset like any other
                const u32 h0 = nr0, h1 = nr1;
                if (m_off == h0 && !src_ll0) {}
                else if (m_off == h1) { nr0 = h1; nr1 = h0; }
                else { nr0 = m_off; nr1 = h0; nr2 = h1; }
                // the windows follow the history: what is kept moves, what is new is loaded
                const u32 k0 = w_r0, k1 = w_r1, k2 = w_r2;
                if (jump_len || ni - (u32)wb > 124u) {
                    wb = (i32)(ni & ~3u);
                    w_cur = load_win(wb);
                    w_r0 = nr0 ? load_win(wb - (i32)nr0) : 0u; w_r1 = nr1 ? load_win(wb - (i32)nr1) : 0u; w_r2 = nr2 ? load_win(wb - (i32)nr2) : 0u;
                } else {
                    auto pick = [&](u32 off) -> u32 { return off == 0 ? 0u : off == r0 ? k0 : off == r1 ? k1 : off == r2 ? k2 : load_win(wb - (i32)off); };
                    w_r0 = pick(nr0); w_r1 = pick(nr1); w_r2 = pick(nr2);
                }
                dirty = nr0 != r0 || nr1 != r1 || nr2 != r2 || jump_len != 0;
                pll_cur = pll0;
            }
            {
                const u64 w = m_len ? ((u64)m_len | ((u64)m_off << 32)) : (0x80000000ull | nlitlen);
                if (ni >= n) end_w = w;
                cur_w = w;
                cur_stored = false;
                if (m_len && ni < n) { if (lane == 0) arrivals[ni] = w; cur_stored = true; }  // (the backward pass steps from match to match: every match's arrival is kept)
            }
            i = ni; price = nprice; litlen = nlitlen; r0 = nr0; r1 = nr1; r2 = nr2;
        }
        // ------------------------------------------------------------------ backward: the sequences, last first, into the end of the chunk's own array
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_sync();
        OptRec* const recs = (OptRec*)seqs;  // (zl_seq and OptRec are both four words)
        const u32 stride = a.seq_stride;
        u32 found = 0, trailing = 0, lit_total = 0;
        bool broken = false;  // (a word that does not step backwards: cannot happen -- and must not hang the device if it ever does: the pass is dropped)
        {
            u32 at = n;
            bool first = true;
            while (at > 0) {
                const u64 w = at == n ? end_w : arrivals[at];
                if ((u32)w & 0x80000000u) {  // a run of literals ends here
                    const u32 k = (u32)w & 0x7FFFFFFFu;
                    if (k == 0 || k > at) { broken = true; break; }
                    if (first) trailing = k;
                    lit_total += k;
                    at -= k;
                    first = false;
                    continue;
                }
                first = false;
                const u32 len = (u32)w & 0xFFFFu, off = (u32)(w >> 32);
                if (len == 0 || len > at) { broken = true; break; }
                const u32 start = at - len;
                u32 k = 0;
                if (start > 0) {
                    const u64 w2 = arrivals[start];
                    if ((u32)w2 & 0x80000000u) k = (u32)w2 & 0x7FFFFFFFu;
                    if (k > start) { broken = true; break; }
                }
                if (found < stride) {
                    if (lane == 0) { OptRec r; r.pos = start; r.len = len; r.off = off; r.litlen = k; recs[stride - 1 - found] = r; }
                    found++;
                    lit_total += k;
                    at = start - k;
                } else {  // (cannot happen: a sequence takes four bytes and the array holds one per four)
                    broken = true;
                    break;
                }
            }
        }
        if (broken) break;  // (the best parse so far stands)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_sync();
        // ------------------------------------------------------------------ emit: turned round, literal counts summed, literals copied
        u32 cum = 0;
        for (u32 b0 = 0; b0 < found; b0 += OPT_THREADS) {
            const u32 k = b0 + lane;
            const bool on = k < found;
            OptRec r; r.pos = r.len = r.off = r.litlen = 0;
            if (on) r = recs[stride - found + k];
            u32 inc = r.litlen;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const u32 up = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += up; }
            const u32 my_cum = cum + inc;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (on) { zl_seq sq; sq.lit_cum = my_cum; sq.match_len = r.len; sq.offset = r.off; sq.pos = r.pos; seqs[k] = sq; }
            // literals: short runs lane by lane, long ones by the whole wave
            const u8* const from = src + r.pos - r.litlen;
            u8* const to = lits + my_cum - r.litlen;
            if (on && r.litlen < 64) for (u32 q = 0; q < r.litlen; q++) to[q] = from[q];
            u64 longm = __ballot(on && r.litlen >= 64);
            while (longm) {
                const int L = __builtin_ctzll(longm);
                longm &= longm - 1;
                const u32 len = (u32)__shfl((int)r.litlen, L, 64), p0 = (u32)__shfl((int)(r.pos - r.litlen), L, 64), d0 = (u32)__shfl((int)(my_cum - r.litlen), L, 64);
                for (u32 q = 16 * lane; q < len; q += 16 * OPT_THREADS) {
                    if (q + 16 <= len) { const uint4 x = ld128(src + p0 + q); __builtin_memcpy(lits + d0 + q, &x, 16); }
                    else for (u32 bq = q; bq < len; bq++) lits[d0 + bq] = src[p0 + bq];
                }
            }
            cum = (u32)__shfl((int)my_cum, 63, 64);
        }
        if (found) {  // the literals behind the last sequence (a chunk without any keeps them in the source)
            const u32 p0 = n - trailing;
            for (u32 q = 16 * lane; q < trailing; q += 16 * OPT_THREADS) {
                if (q + 16 <= trailing) { const uint4 x = ld128(src + p0 + q); __builtin_memcpy(lits + cum + q, &x, 16); }
                else for (u32 bq = q; bq < trailing; bq++) lits[cum + bq] = src[p0 + bq];
            }
        }
        nseq = found;
        nlit = lit_total;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_sync();
        // ------------------------------------------------------------------ is it better?  (Where a pass does not improve on the parse whose statistics priced
        // it, the next would be the same pass again.  It happens: i.i.d. symbols over a small alphabet -- every position has a five-byte twin, matches at
        // 15 bits look as good as four literals, a parse made of them prices every literal RUN out of reach, and the result is 4 % larger than no match at all.)
        const float cost = take_stats(lits, seqs, nseq, nlit);
        if (!(cost < best_cost)) break;
        best_cost = cost; best_buf ^= 1; best_nseq = nseq; best_nlit = nlit;
    }
    if (best_buf == 1) {  // the entropy stage reads the launch's own arrays
        const uint4* const qs = (const uint4*)seq_buf[1];
        uint4* const qd = (uint4*)seq_buf[0];
        for (u32 k = lane; k < best_nseq; k += OPT_THREADS) qd[k] = qs[k];
        if (best_nseq) for (u32 q = 16 * lane; q < best_nlit; q += 16 * OPT_THREADS) {
            if (q + 16 <= best_nlit) { const uint4 x = ld128(lit_buf[1] + q); __builtin_memcpy(lit_buf[0] + q, &x, 16); }
            else for (u32 bq = q; bq < best_nlit; bq++) lit_buf[0][bq] = lit_buf[1][bq];
        }
    }
    if (lane == 0) { a.meta[c].nseq = best_nseq; a.meta[c].nlit = best_nlit; }
}

// a match longer than its measure arrives as pieces with one offset and no literal between them: joined here (zl_match_common.h)
__global__ __launch_bounds__(256) void zl_opt_merge_kernel(zl_match_args a) {
    __shared__ u32 t_end[256], t_off[256], scan_a[4];
    const u32 c = blockIdx.x;
    zl_seq* const seqs = a.seqs + (size_t)c * a.seq_stride;
    const u32 out_n = zl_merge_pieces(seqs, a.meta[c].nseq, t_end, t_off, scan_a);
    if (threadIdx.x == 0) a.meta[c].nseq = out_n;
}

extern "C" hipError_t zl_launch_opt_parse(const zl_match_args* a, const u64* chain, u64* arrivals, u32 nchunks, u32 passes, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(zl_opt_parse_kernel, dim3(nchunks), dim3(OPT_THREADS), 0, stream, *a, chain, arrivals, nchunks, passes);
    hipLaunchKernelGGL(zl_opt_merge_kernel, dim3(nchunks), dim3(256), 0, stream, *a);
    return hipGetLastError();
}
