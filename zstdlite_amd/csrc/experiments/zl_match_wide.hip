// zl_match_wide.hip -- E1 for the bulk path (no dictionary, independent frames): the LZ77 parse of a 128 KiB chunk
// by ONE 1024-thread workgroup per CU with the whole search window resident in LDS.
//
// The 256-thread kernel of zl_encode.hip reads its candidates from global memory: three random 8-byte gathers per
// position, one line request per lane -- 8 GB of traffic per GiB of input (profiles/r02_pmc_summary.json) and a texture
// path that is busy for ~770 cycles per 256-position window, which six workgroups per CU can hide but not shorten.
// gfx950 has 160 KiB of LDS per CU: enough for the last 64 KiB of the chunk (a ring), both match tables with full
// 32-bit positions, and the parse arrays of a window of 1024 positions.  Every candidate lies at most 60 KiB back, so
// EVERY candidate read is an LDS read (three aligned dwords and two v_alignbyte); global memory is read once, in order,
// a window ahead, and written once (literals, sequences).  Same algorithm as the 256-thread kernel -- the double-hash
// greedy parse of ZSTD_compressBlock_doubleFast (zstd.c:29909-30116) window by window, stride 1 where matches are dense,
// stride 2 where they are not -- with these differences:
//   * a window is 1024 positions (2048 bytes in the stride-2 body): the tables know positions in front of the window
//     only, so repeats at a distance below the window length are seen later than by the 256-position kernel
//     (scripts/dev/sim/sim.c: text -0.7 % ratio);
//   * inserts are LDS atomicMax on (position + 1): the most recent position wins whatever the order of the waves, so no
//     wave-after-wave insert phases (determinism);
//   * the 16 wave segments of the parse are chained with v_readlane on rows held in registers (each wave reads all 16
//     rows of exit points with independent LDS reads, then follows the one path through them), not with dependent LDS
//     reads;
//   * the one-step lazy rule looks at the neighbouring lane with a shuffle; the last lane of a wave does not look.
#include <hip/hip_runtime.h>

#include "zl_device.h"
#include "zl_encode.h"
#include "zl_kernels.h"
#include "zl_match_common.h"

#define WT 1024u                 // threads per workgroup
#define WW 16u                   // waves
#define WWIN 1024u               // positions per window, stride-1 body
#define WWIN2 2048u              // bytes per window, stride-2 body
#define RING_BYTES 65536u
#define RING_PAD 16u             // the first bytes again behind the end: a 12-byte read never wraps
#define RING_AHEAD 4096u         // the ring holds the chunk up to this far behind the start of the window being searched
#define RING_REACH (RING_BYTES - RING_AHEAD)  // a position this far in front of the window's start is still resident
#ifndef WLONG_LOG
#define WLONG_LOG 13
#endif
#ifndef WSHORT_LOG
#define WSHORT_LOG 12
#endif
#ifndef ZL_WDENSE_SEQS
#define ZL_WDENSE_SEQS 96        // sequences per 2048 bytes from which on the stride-1 body is used (24 per 512, as the 256-thread kernel)
#endif
#ifndef ZL_WLAZY_BIAS
#define ZL_WLAZY_BIAS 8
#endif
#ifndef ZL_WSPARSE_MIN
#define ZL_WSPARSE_MIN 5
#endif
#ifndef ZL_WSKIP_MAX
#define ZL_WSKIP_MAX 7           // at most this many windows are copied unsearched behind a searched one
#endif
#ifndef ZL_WMISS_STEP
#define ZL_WMISS_STEP 2          // a window without a sequence counts this much; every 4 counts add one skipped window
#endif

__device__ __forceinline__ u32 whash8(u64 v) { return (u32)((v * 0x9E3779B185EBCA87ULL) >> (64 - WLONG_LOG)); }
__device__ __forceinline__ u32 whash6(u64 v) { return (u32)(((v << (64 - 8 * ZL_SHORT_BYTES)) * 0xC2B2AE3D27D4EB4FULL) >> (64 - WSHORT_LOG)); }

struct WideShared {
    u32 ring[(RING_BYTES + RING_PAD) / 4];  // byte q of the chunk lives at q mod 64 KiB
    u32 tbl_long[1u << WLONG_LOG];          // position + 1 of the most recent insert, 0 = empty
    u32 tbl_short[1u << WSHORT_LOG];
    u32 cand[WT];                           // stride-2 body: backward-extension requests
    u16 jump[2][WT];
    u8 mark[WT];
    u32 scan_a[WW], scan_b[WW];
    u32 rep;                                // offset of the most recent sequence (repeat-offset candidate)
};

__device__ __forceinline__ u64 ring_ld64(const u32* ring, u32 pos) {
    const u32 a = pos & (RING_BYTES - 1);
    const u32* w = ring + (a >> 2);
    const u32 w0 = w[0], w1 = w[1], w2 = w[2];
    const u32 lo = __builtin_amdgcn_alignbyte(w1, w0, a & 3u), hi = __builtin_amdgcn_alignbyte(w2, w1, a & 3u);
    return (u64)lo | ((u64)hi << 32);
}
__device__ __forceinline__ u32 ring_ld8(const u32* ring, u32 pos) { return ((const u8*)ring)[pos & (RING_BYTES - 1)]; }
__device__ __forceinline__ void ring_st64(u32* ring, u32 pos /* multiple of 8 */, u64 v) {
    const u32 a = pos & (RING_BYTES - 1);
    *(u64*)((u8*)ring + a) = v;
    if (a < RING_PAD) *(u64*)((u8*)ring + RING_BYTES + a) = v;
}
__device__ __forceinline__ u64 src_ld64_clamped(const u8* src, u32 pos, u32 n) {  // bytes behind the chunk's end read as zero
    if (pos + 8 <= n) return ld64(src + pos);
    u64 v = 0;
    for (u32 i = 0; i < 8; i++) if (pos + i < n) v |= (u64)src[pos + i] << (8 * i);
    return v;
}
__device__ __forceinline__ u32 uniform(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }

// The one path through the window's 16 wave segments.  rows[k] holds, lane by lane, where a path that enters segment k
// at that lane leaves it (an absolute node number >= 64 (k + 1)); the path enters the window at `enter`.  Returns the
// node at which the path leaves the window; my_entry = where it enters this wave's segment (or none).
__device__ __forceinline__ u32 chain_segments(const u16* jump, u32 enter, u32 wave, u32 lane, u32& my_entry) {
    u32 rows[WW];
#pragma unroll
    for (u32 k = 0; k < WW; k++) rows[k] = jump[64 * k + lane];
    my_entry = 0xFFFFFFFFu;
    enter = uniform(enter);
#pragma unroll
    for (u32 k = 0; k < WW; k++) {
        if (enter < 64 * (k + 1)) {
            if (k == wave) my_entry = enter;
            enter = (u32)__builtin_amdgcn_readlane((int)rows[k], (int)(enter - 64 * k));
        }
    }
    return enter;
}

// Totals of the 16 per-wave counts (two 16-bit fields per word) and the sum of those in front of `wave`.
__device__ __forceinline__ void wave_counts(const u32* scan, u32 wave, u32& pre, u32& tot) {
    pre = 0; tot = 0;
#pragma unroll
    for (u32 w4 = 0; w4 < WW; w4 += 4) {
        const uint4 x = *(const uint4*)(scan + w4);
        const u32 xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (u32 j = 0; j < 4; j++) {
            if (w4 + j < wave) pre += xs[j];
            tot += xs[j];
        }
    }
}

template <bool PROF>
__global__ __launch_bounds__(WT) void zl_match_wide_kernel(zl_match_args a, u32 nchunks) {
    __shared__ __attribute__((aligned(16))) WideShared S;
    ZL_POISON(S, 4);
    const u32 t = threadIdx.x, lane = t & 63u, wave = uniform(t >> 6), wbase = t & ~63u;
    const u32 c = zl_spread(blockIdx.x, nchunks);
    const zl_chunk_desc cd = a.chunks[c];
    const u8* src = a.src + cd.src_off;
    const u32 n = cd.src_size;
    u8* lits = a.lits + (size_t)c * a.lit_stride;
    zl_seq* seqs = a.seqs + (size_t)c * a.seq_stride;

    for (u32 i = t; i < (1u << WLONG_LOG); i += WT) S.tbl_long[i] = 0u;
    for (u32 i = t; i < (1u << WSHORT_LOG); i += WT) S.tbl_short[i] = 0u;
    if (t == 0) S.rep = 0;
    if (8 * t < RING_AHEAD && 8 * t < n) ring_st64(S.ring, 8 * t, src_ld64_clamped(src, 8 * t, n));
    lds_barrier();

    u32 nseq = 0, nlit = 0;  // uniform running totals
    u32 miss = 0;            // consecutive windows without a sequence (uniform)
    u32 carry = 0;           // where the parse enters the next window, in bytes (uniform)
    bool dense = !a.sparse_only;  // which body the next window gets (uniform)
    const bool prof = PROF && a.stats != nullptr && t == 0;
    u64 c_find = 0, c_parse = 0, c_emit = 0, c_ins = 0, c_it = 0;
    for (u32 ip = 0; ip < n;) {
        const u32 win = dense ? WWIN : WWIN2;
        u32 win_seqs = 0;
        const u64 s0 = prof ? __builtin_amdgcn_s_memtime() : 0;
        u64 s1 = s0, s2 = s0, s3 = s0;
        // the piece of the chunk that enters the ring at the end of this iteration is fetched now
        const u32 pq = ip + RING_AHEAD + 8 * t;
        const bool pf = 8 * t < win && pq < n;
        u64 pv = 0;
        if (pf) pv = src_ld64_clamped(src, pq, n);
        const u32 rep = S.rep;
        const int resident = (int)ip - (int)RING_REACH;  // positions from here on are in the ring
        if (dense) {
            const u32 p = ip + t;
            const bool in_range = p < n, can_match = p + 8 <= n;  // the last 7 bytes of a chunk are always literals
            const u64 v = ring_ld64(S.ring, p);
            const bool search = carry < WWIN;  // otherwise the whole window lies inside an earlier match
            u32 hl = 0, hs = 0, best_len = 0, best_off = 0;
            if (can_match) {
                hl = whash8(v);
                hs = whash6(v);
                if (search) {
                    const u32 el = S.tbl_long[hl], es = S.tbl_short[hs];
                    const bool have_r = rep != 0 && p >= rep && (int)(p - rep) >= resident;
                    const bool have_l = el != 0 && (int)(el - 1) >= resident;
                    const bool have_s = es != 0 && (int)(es - 1) >= resident && es != el;
                    const u32 q_r = have_r ? p - rep : p, q_l = have_l ? el - 1 : p, q_s = have_s ? es - 1 : p;
                    const u64 w_r = ring_ld64(S.ring, q_r), w_l = ring_ld64(S.ring, q_l), w_s = ring_ld64(S.ring, q_s);
                    const u32 l_r = common_bytes(v, w_r), l_l = common_bytes(v, w_l), l_s = common_bytes(v, w_s);
                    if (have_r && l_r >= MIN_MATCH) { best_len = l_r; best_off = rep; }
                    if (have_l && l_l >= MIN_MATCH + 1 && l_l > best_len) { best_len = l_l; best_off = p - q_l; }
                    if (have_s && best_len < 8 && l_s >= MIN_MATCH + 1 && l_s > best_len) { best_len = l_s; best_off = p - q_s; }
                    // extend while whole 8-byte words keep matching
                    while (best_len >= 8 && (best_len & 7) == 0 && best_len < LANE_MATCH_CAP && p + best_len + 8 <= n) {
                        const u32 l = common_bytes(ring_ld64(S.ring, p + best_len), ring_ld64(S.ring, p - best_off + best_len));
                        best_len += l;
                        if (l < 8) break;
                    }
                }
            }
            bool has_match = best_len >= MIN_MATCH;
            // ---- one-step lazy evaluation (cf. the ip+1 long-match check of zstd.c:30025-30040): if the next position
            // holds a clearly better match, emit this byte as a literal instead
            {
                const u32 mine = has_match ? (best_len | (best_off << 8)) : 0u;
                const u32 nx = __shfl_down(mine, 1, 64);
                if (has_match && lane != 63 && nx) {
                    const int g1 = 4 * (int)best_len - (int)(31 - __clz(best_off + 1)) + 4;
                    const int g2 = 4 * (int)(nx & 0xFF) - (int)(31 - __clz((nx >> 8) + 1));
                    if (g2 > g1) has_match = false;
                }
            }
            if (prof) s1 = __builtin_amdgcn_s_memtime();
            // ---- greedy parse over the window: every wave closes its 64 (position -> next position) links by pointer
            // jumping through lane shuffles, which tells for every entry point where the path leaves the segment
            const u32 nxt = t + (has_match ? best_len : 1u);
            u32 jr[6];
            u32 jp = nxt;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                jr[r] = jp;
                const u32 far = __shfl(jp, (int)((jp - wbase) & 63), 64);
                jp = (jp < wbase + 64) ? far : jp;
            }
            S.jump[0][t] = (u16)jp;
            lds_barrier();
            u32 my_entry;
            const u32 leave = chain_segments(S.jump[0], carry, wave, lane, my_entry);
            const u32 next_carry = leave - WWIN;  // uniform: where the path enters the next window
            // (marks stay inside the wave's 64 bytes: LDS executes a wave's accesses in program order, the asm
            // statements only keep the compiler from caching or reordering them)
            S.mark[t] = (t == my_entry) ? 1 : 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                asm volatile("" ::: "memory");
                if (S.mark[t] && jr[r] < wbase + 64) S.mark[jr[r]] = 1;
            }
            asm volatile("" ::: "memory");
            const bool on_path = S.mark[t] != 0;
            const bool is_seq = on_path && has_match, is_lit = on_path && !has_match && in_range;
            if (prof) s2 = __builtin_amdgcn_s_memtime();
            const u64 ms = __ballot(is_seq), ml = __ballot(is_lit);
            const u64 below = (1ull << lane) - 1;
            if (lane == 0) S.scan_a[wave] = (u32)__popcll(ms) | ((u32)__popcll(ml) << 16);
            lds_barrier();
            u32 pre, tot;
            wave_counts(S.scan_a, wave, pre, tot);
            const u32 ex_s = (pre & 0xFFFF) + (u32)__popcll(ms & below), ex_l = (pre >> 16) + (u32)__popcll(ml & below);
            const u32 tot_s = tot & 0xFFFF, tot_l = tot >> 16;
            if (is_lit) lits[nlit + ex_l] = (u8)v;
            if (is_seq) {
                zl_seq sq;
                sq.lit_cum = nlit + ex_l;
                sq.match_len = best_len;
                sq.offset = best_off;
                sq.pos = p;
                seqs[nseq + ex_s] = sq;
                if (ex_s + 1 == tot_s) S.rep = best_off;  // last sequence of the window
            }
            carry = next_carry;
            if (prof) s3 = __builtin_amdgcn_s_memtime();
            // ---- insert every position of the window: the most recent position wins (atomicMax on position + 1)
            if (can_match) {
                atomicMax(&S.tbl_long[hl], p + 1);
                atomicMax(&S.tbl_short[hs], p + 1);
            }
            nseq += tot_s;
            nlit += tot_l;
            win_seqs = tot_s;
        } else {
            // the stride-2 parse runs over even positions ("nodes"): an odd entry point is one literal first
            const bool covered = carry >= WWIN2;
            if (!covered && (carry & 1)) {
                if (ip + carry < n) {
                    if (t == 0) lits[nlit] = (u8)ring_ld8(S.ring, ip + carry);
                    nlit += 1;
                }
                carry += 1;
            }
            const u32 carry_n = carry >> 1;
            const u32 p = ip + 2 * t;
            const bool in0 = p < n, in1 = p + 1 < n;
            const bool can0 = p + 8 <= n, can1 = p + 9 <= n;  // the last 7 bytes of a chunk are always literals
            // 12 bytes from p (p is even): the 8 bytes at p, the 8 at p + 1; and the byte in front of p
            u64 v0, v1;
            u32 byte_before;
            {
                const u32 ar = p & (RING_BYTES - 1);
                const u32* w = S.ring + (ar >> 2);
                const u32 w0 = w[0], w1 = w[1], w2 = w[2];
                const u32 sh = ar & 3u;  // 0 or 2
                v0 = (u64)__builtin_amdgcn_alignbyte(w1, w0, sh) | ((u64)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32);
                v1 = (u64)__builtin_amdgcn_alignbyte(w1, w0, sh + 1) | ((u64)__builtin_amdgcn_alignbyte(w2, w1, sh + 1) << 32);
                byte_before = p ? ring_ld8(S.ring, p - 1) : 0x100u;
            }
            const bool search = carry_n < WT;  // otherwise the whole window lies inside an earlier match
            u32 hl0 = 0, hs0 = 0, hl1 = 0, hs1 = 0;
            u32 best_len = 0, best_off = 0;
            if (can0) {
                hl0 = whash8(v0); hs0 = whash6(v0);
                if (search) {
                    const u32 el = S.tbl_long[hl0], es = S.tbl_short[hs0];
                    const bool have_r = rep != 0 && p >= rep && (int)(p - rep) >= resident;
                    const bool have_l = el != 0 && (int)(el - 1) >= resident;
                    const bool have_s = es != 0 && (int)(es - 1) >= resident && es != el;
                    const u32 q_r = have_r ? p - rep : p, q_l = have_l ? el - 1 : p, q_s = have_s ? es - 1 : p;
                    const u64 w_r = ring_ld64(S.ring, q_r), w_l = ring_ld64(S.ring, q_l), w_s = ring_ld64(S.ring, q_s);
                    const u32 l_r = common_bytes(v0, w_r), l_l = common_bytes(v0, w_l), l_s = common_bytes(v0, w_s);
                    if (have_r && l_r >= MIN_MATCH) { best_len = l_r; best_off = rep; }
                    // (a 4-byte hit is kept for now: it is worth a new offset only if the byte in front matches too, see below)
                    if (have_l && l_l >= MIN_MATCH && l_l > best_len) { best_len = l_l; best_off = p - q_l; }
                    if (have_s && best_len < 8 && l_s >= MIN_MATCH && l_s > best_len) { best_len = l_s; best_off = p - q_s; }
                    while (best_len >= 8 && (best_len & 7) == 0 && best_len < LANE_MATCH_CAP && p + best_len + 8 <= n) {
                        const u32 l = common_bytes(ring_ld64(S.ring, p + best_len), ring_ld64(S.ring, p - best_off + best_len));
                        best_len += l;
                        if (l < 8) break;
                    }
                }
            }
            if (can1) { hl1 = whash8(v1); hs1 = whash6(v1); }  // the odd position is only inserted
            bool has_match = best_len >= MIN_MATCH;
            // A repeat that began on the odd position in front of p is found here one byte late: if the byte in front
            // of the candidate equals the byte in front of p and that byte is still a free literal, the match starts there.
            u32 cand_before = 0x200;
            if (has_match && p >= 1 && p - best_off >= 1 && (int)(p - best_off - 1) >= resident) cand_before = ring_ld8(S.ring, p - best_off - 1);
            const bool back_ok = cand_before == byte_before;
            if (has_match && best_off != rep && best_len + (back_ok ? 1u : 0u) < ZL_WSPARSE_MIN) has_match = false;
            // lazy evaluation against the next probed position (two bytes on): giving up this match costs two literals
            {
                const u32 mine = has_match ? (best_len | (best_off << 8)) : 0u;
                const u32 nxc = __shfl_down(mine, 1, 64);
                if (has_match && lane != 63 && nxc) {
                    const int g1 = 4 * (int)best_len - (int)(31 - __clz(best_off + 1)) + ZL_WLAZY_BIAS;
                    const int g2 = 4 * (int)(nxc & 0xFF) - (int)(31 - __clz((nxc >> 8) + 1));
                    if (g2 > g1) has_match = false;
                }
            }
            // does any lane of the workgroup hold a match?
            {
                const u64 many = __ballot(has_match);
                if (lane == 0) S.scan_b[wave] = many ? 1u : 0u;
            }
            if (prof) { s1 = __builtin_amdgcn_s_memtime(); s2 = s1; }
            lds_barrier();
            u32 any_pre, any_tot;
            wave_counts(S.scan_b, 0, any_pre, any_tot);
            const bool any_match = uniform(any_tot) != 0;
            u32 tot_s = 0, tot_l = 0;
            if (!any_match) {
                // ---- a window without a single match -- the rule on data that does not repeat -- needs no parse: every node
                // from the entry point on is on the path and gives its two bytes (what is left of them at the end of the chunk)
                if (!covered && t >= carry_n) {
                    const u32 ex = 2 * (t - carry_n);
                    if (in0) lits[nlit + ex] = (u8)v0;
                    if (in1) lits[nlit + ex + 1] = (u8)(v0 >> 8);
                }
                if (!covered && ip + 2 * carry_n < n) {
                    const u32 left = n - (ip + 2 * carry_n), span = 2 * (WT - carry_n);
                    tot_l = left < span ? left : span;
                }
                carry = covered ? carry - WWIN2 : 0u;
            } else {
                // the byte behind a match that ends on an odd position is a literal of this lane
                const u32 mend = 2 * t + best_len;  // window-relative end of the match
                const bool odd_end = has_match && (mend & 1) && p + best_len < n;
                u32 tail_byte = 0;
                if (odd_end) tail_byte = ring_ld8(S.ring, p + best_len);
                // ---- greedy parse over the 1024 nodes (even positions): wave-local closure, chain, marks
                const u32 nxt = has_match ? ((mend + 1) >> 1) : t + 1u;  // next node on the path
                u32 jr[6];
                u32 jp = nxt;
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    jr[r] = jp;
                    const u32 far = __shfl(jp, (int)((jp - wbase) & 63), 64);
                    jp = (jp < wbase + 64) ? far : jp;
                }
                S.jump[0][t] = (u16)jp;  // where a path entering at node t leaves this wave's segment
                lds_barrier();
                u32 my_entry;
                const u32 leave = chain_segments(S.jump[0], carry_n, wave, lane, my_entry);
                const u32 next_carry = leave - WT;
                S.mark[t] = (t == my_entry) ? 1 : 0;
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    asm volatile("" ::: "memory");
                    if (S.mark[t] && jr[r] < wbase + 64) S.mark[jr[r]] = 1;
                }
                asm volatile("" ::: "memory");
                if (prof) s2 = __builtin_amdgcn_s_memtime();
                const bool on_path = S.mark[t] != 0;
                const bool is_seq = on_path && has_match;
                // backward extension by one byte.  The byte in front of an on-path node is a literal owned by the node
                // before it on the path: the odd byte of a lane without a match, or the byte behind a match that ends on
                // an odd position.  The wanting lane raises a flag in its slot, the owner answers in the second array;
                // both then count one literal less / one match byte more.
                const bool givable = on_path && nxt < WT && (has_match ? odd_end : in1);
                S.cand[t] = (is_seq && back_ok) ? 1u : 0u;
                S.jump[1][t] = 0;
                lds_barrier();
                const bool give = givable && S.cand[nxt] != 0;
                if (give) S.jump[1][nxt] = 1;
                lds_barrier();
                const bool ext = S.jump[1][t] != 0;
                const u32 nl_base = !on_path ? 0u : (has_match ? (odd_end ? 1u : 0u) : (in1 ? 2u : (in0 ? 1u : 0u)));
                const u32 nl_mine = nl_base - (give ? 1u : 0u);
                // compaction ranks from ballots (literal counts are 0..2: two bit planes)
                const u64 ms = __ballot(is_seq), m1 = __ballot((nl_mine & 1) != 0), m2 = __ballot((nl_mine & 2) != 0);
                const u64 below = (1ull << lane) - 1;
                if (lane == 0) S.scan_a[wave] = (u32)__popcll(ms) | (((u32)__popcll(m1) + 2 * (u32)__popcll(m2)) << 16);
                lds_barrier();
                u32 pre, tot;
                wave_counts(S.scan_a, wave, pre, tot);
                const u32 ex_s = (pre & 0xFFFF) + (u32)__popcll(ms & below);
                const u32 ex_l = (pre >> 16) + (u32)__popcll(m1 & below) + 2 * (u32)__popcll(m2 & below);
                tot_s = tot & 0xFFFF;
                tot_l = tot >> 16;
                if (is_seq) {
                    zl_seq sq;
                    sq.lit_cum = nlit + ex_l;
                    sq.match_len = best_len + (ext ? 1u : 0u);
                    sq.offset = best_off;
                    sq.pos = p - (ext ? 1u : 0u);
                    seqs[nseq + ex_s] = sq;
                    if (ex_s + 1 == tot_s) S.rep = best_off;  // last sequence of the window
                    if (odd_end && !give) lits[nlit + ex_l] = (u8)tail_byte;
                } else if (nl_mine) {
                    lits[nlit + ex_l] = (u8)v0;
                    if (nl_mine == 2) lits[nlit + ex_l + 1] = (u8)(v0 >> 8);
                }
                carry = covered ? carry - WWIN2 : 2 * next_carry;
            }
            if (prof) s3 = __builtin_amdgcn_s_memtime();
            // ---- insert both positions of every lane
            if (can0) { atomicMax(&S.tbl_long[hl0], p + 1); atomicMax(&S.tbl_short[hs0], p + 1); }
            if (can1) { atomicMax(&S.tbl_long[hl1], p + 2); atomicMax(&S.tbl_short[hs1], p + 2); }
            nseq += tot_s;
            nlit += tot_l;
            win_seqs = tot_s;
        }
        if (pf) ring_st64(S.ring, pq, pv);  // (nothing reads the ring between here and the barrier)
        lds_barrier();
        if (prof) {
            const u64 s4 = __builtin_amdgcn_s_memtime();
            c_find += s1 - s0; c_parse += s2 - s1; c_emit += s3 - s2; c_ins += s4 - s3; c_it++;
        }
        // ---- search acceleration on data without matches (cf. the growing step of zstd.c:30018-30023): after a run of
        // windows that produced no sequence, the next windows are emitted as literals without searching or inserting.  A
        // pure function of the data, so the output stays deterministic.  Any sequence resets it.
        miss = (win_seqs == 0 && carry == 0) ? miss + ZL_WMISS_STEP : 0;
        u32 skip = miss >> 2;
        if (skip > ZL_WSKIP_MAX) skip = ZL_WSKIP_MAX;
        ip += win;
        if (skip && ip < n) {
            const u32 to = (ip + skip * win < n) ? ip + skip * win : n;
            {   // 16 bytes per thread and trip (global memory takes unaligned vectors)
                const u32 cnt = to - ip, body = cnt & ~15u;
                for (u32 q = t * 16; q < body; q += WT * 16) { uint4 x; __builtin_memcpy(&x, src + ip + q, 16); __builtin_memcpy(lits + nlit + q, &x, 16); }
                if (t < cnt - body) lits[nlit + body + t] = src[ip + body + t];
            }
            // the ring follows: what the skipped windows would have brought in
            for (u32 q = ip + RING_AHEAD + 8 * t; q < ip + skip * win + RING_AHEAD && q < n; q += 8 * WT) ring_st64(S.ring, q, src_ld64_clamped(src, q, n));
            lds_barrier();
            nlit += to - ip;
            ip = (ip + skip * win < n) ? ip + skip * win : n;
        }
        dense = !a.sparse_only && win_seqs * (WWIN2 / win) >= ZL_WDENSE_SEQS;
    }
    if (prof) {
        atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)c_find);
        atomicAdd((unsigned long long*)&a.stats[1], (unsigned long long)c_parse);
        atomicAdd((unsigned long long*)&a.stats[2], (unsigned long long)c_emit);
        atomicAdd((unsigned long long*)&a.stats[3], (unsigned long long)c_ins);
        atomicAdd((unsigned long long*)&a.stats[4], (unsigned long long)c_it);
        atomicAdd((unsigned long long*)&a.stats[5], (unsigned long long)nseq);
    }
    // ---- merge continuation pieces (as the 256-thread kernel does: a lane verifies at most LANE_MATCH_CAP bytes, so a
    // longer match arrives as several sequences without a literal between them; joined, they are the one sequence the
    // reference would have found).  One pass over the chunk's sequences, 1024 at a time, compacting in place.
    u32 out_n = nseq;
    if (nseq > 1) {
        zl_sync_global();  // the sequences were written by other lanes
        u32* const t_end = S.tbl_long;       // [1024] where the sequence of lane t ends
        u32* const t_off = S.tbl_long + WT;  // [1024] its offset
        u32 carried_end = 0xFFFFFFFFu, carried_off = 0;  // of the last sequence of the tile before (uniform)
        out_n = 0;
        for (u32 base = 0; base < nseq; base += WT) {
            const u32 i = base + t;
            const bool on = i < nseq;
            zl_seq sq;
            sq.lit_cum = sq.match_len = sq.offset = sq.pos = 0;
            if (on) sq = seqs[i];
            t_end[t] = on ? sq.pos + sq.match_len : 0xFFFFFFFFu;
            t_off[t] = sq.offset;
            zl_sync_global();  // every lane holds its sequence before any slot of the array is rewritten
            const u32 pe = t ? t_end[t - 1] : carried_end, po = t ? t_off[t - 1] : carried_off;
            const bool cont = on && sq.pos == pe && sq.offset == po;
            const bool head = on && !cont;
            const u64 mh = __ballot(head);
            if (lane == 0) S.scan_a[wave] = (u32)__popcll(mh);
            const u32 last = (nseq - base < WT ? nseq - base : WT) - 1;
            const u32 next_end = t_end[last], next_off = t_off[last];
            lds_barrier();
            u32 pre, tot;
            wave_counts(S.scan_a, wave, pre, tot);
            // heads up to and including this lane: the head's new slot is that count minus one
            const u32 dest = out_n + pre + (u32)__popcll(mh & ((2ull << lane) - 1)) - 1;
            if (head) seqs[dest] = sq;
            zl_sync_global();  // heads are in place before their pieces add to them
            if (cont) atomicAdd(&seqs[dest].match_len, sq.match_len);
            carried_end = next_end;
            carried_off = next_off;
            out_n += tot;
            lds_barrier();
        }
    }
    if (t == 0) {
        a.meta[c].nseq = out_n;
        a.meta[c].nlit = nlit;
    }
}

extern "C" hipError_t zl_launch_match_wide(const zl_match_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    if (a->stats) hipLaunchKernelGGL((zl_match_wide_kernel<true>), dim3(nchunks), dim3(WT), 0, stream, *a, nchunks);
    else hipLaunchKernelGGL((zl_match_wide_kernel<false>), dim3(nchunks), dim3(WT), 0, stream, *a, nchunks);
    return hipGetLastError();
}
