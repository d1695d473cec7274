// zl_encode.hip -- chunk-parallel zstd encoder for gfx950 (MI355X).
//
// The input is cut into independent 128 KiB chunks; each chunk becomes one single-segment zstd frame
// of up to eight blocks that share one set of entropy tables (the reference's
// ZSTD_compress_frameChunk, zstd.c:27007, walks 128 KiB blocks serially inside one frame).  Kernels:
//
//   zl_match_auto_kernel  E1: LZ77 parse, one 256-thread workgroup per chunk.  Restates the double-hash
//                      greedy parse of ZSTD_compressBlock_doubleFast (zstd.c:29909-30116) for a wide
//                      machine: each iteration looks at a window of 256 (or, where matches are sparse,
//                      512) consecutive positions at once -- every lane hashes its position with an
//                      8-byte and a 6-byte multiplicative hash, probes two LDS tables of 16-bit entries
//                      holding the most recent earlier position per bucket, verifies and extends its
//                      best candidate (plus the running repeat offset); the greedy parse over the
//                      (position -> position + matchLength | +1) links is closed per wave with lane
//                      shuffles and chained across the four waves through LDS instead of a serial walk;
//                      on-path matches become sequences, on-path non-matches literals (ranks from
//                      ballots), and the window's positions are inserted wave after wave with plain
//                      stores, so that the table content -- and therefore the output -- does not
//                      depend on wave scheduling (determinism).
//   zl_entropy_kernel  E2: per chunk, Huffman-code the literals (4 streams per block, one wave each)
//                      and FSE-code the sequences (zstd.c:20692 ZSTD_compressLiterals, zstd.c:25265
//                      ZSTD_buildSequencesStatistics, zstd.c:21146 ZSTD_encodeSequences): histograms in
//                      LDS, the Huffman tree on one wave next to the three FSE tables on three others,
//                      the FSE state chains of all blocks side by side on the lanes of one wave, and
//                      all bit placement by prefix scan + LDS atomicOr so that stores are whole words.
//                      Raw / RLE fallbacks as ZSTD_compressBlock_internal (zstd.c:26833).  With a
//                      dictionary, single-block frames may take its tables (zl_cdict_build_kernel).
//   zl_scan_sizes_kernel, zl_gather_kernel   frames are produced at a fixed stride, then packed back to
//                      back after an exclusive scan of their sizes.
#include <hip/hip_runtime.h>

#include "zl_device.h"
#include "zl_encode.h"
#include "zl_format.h"
#include "zl_kernels.h"

#define ENC_THREADS 256
#define PPT 1              // positions per thread and iteration (2 was measured: same speed, 0.8 % worse ratio)
#define WIN (ENC_THREADS * PPT)  // positions examined per iteration
#define WIN_LOG (PPT == 1 ? 8 : 9)
#include "zl_match_common.h"

#define ZL_WBUF_BYTES 352  // 256 positions of a window + the longest verification (72) + a last word and its alignment slack
template <int LLOG, int SLOG, bool WBUF = false>
struct MatchSharedT {
    // 16-bit entries, two per word: (position + 1) mod 2^16, 0 = empty.  A lookup takes the entry to
    // mean the nearest such position in front of it, so candidates reach 64 KiB back (an older entry
    // aliases to a wrong position and fails verification).  Half the LDS of 32-bit entries: six
    // workgroups per CU instead of three, and this kernel is latency-bound.
    u32 tbl_long[(1 << LLOG) / 2];
    u32 tbl_short[(1 << SLOG) / 2];
    u16 jump[2][WIN];
    u32 cand[WIN];  // [7:0] match length, [31:8] offset of each lane's best match
    u32 cand_end;   // always 0: what the window's last lane reads as its neighbour's candidate (lean body)
    u32 scan_a[4], scan_b[4];
    u32 rep;      // offset of the most recent sequence (repeat-offset candidate)
    // lean body: the window's own bytes [wbuf_ip, + 320): every lane's eight and the words the extension loop compares them
    // with.  Twenty lanes fetch them (16 bytes each) where 256 lanes fetched eight overlapping bytes each and every extension
    // round 64 more: a lane-load is a tag look-up in the CU's L1 whatever its neighbours read, and those look-ups bound the kernel.
    alignas(16) u8 wbuf[WBUF ? ZL_WBUF_BYTES : 16];
};

__device__ __forceinline__ void block_scan2(u32* sa, u32* sb, u32 a, u32 b, u32& ex_a, u32& ex_b, u32& tot_a, u32& tot_b) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 ta = __shfl_up(ia, d, 64), tb = __shfl_up(ib, d, 64);
        if ((int)lane >= d) { ia += ta; ib += tb; }
    }
    if (lane == 63) { sa[wave] = ia; sb[wave] = ib; }
    __syncthreads();
    u32 pa = 0, pb = 0, ta = 0, tb = 0;
#pragma unroll
    for (u32 w = 0; w < 4; w++) {
        const u32 xa = sa[w], xb = sb[w];
        if (w < wave) { pa += xa; pb += xb; }
        ta += xa; tb += xb;
    }
    ex_a = pa + ia - a;
    ex_b = pb + ib - b;
    tot_a = ta;
    tot_b = tb;
    __syncthreads();
}

// Deterministic insert of position p (inside the window that starts at ip) into a 16-bit entry:
// the most recent position of the window wins, whatever was there before loses.  `old` is the word
// as this lane saw it during the lookup; in the common case one compare-and-swap per table settles it.
__device__ __forceinline__ bool insert16_wants(u32 old, u32 sh, u32 mine, u32 ip, u32 win) {
    const u32 cur = (old >> sh) & 0xFFFF;
    const u32 age = (cur - 1 - ip) & 0xFFFF;  // < win: written by a position of this window
    return !(cur != 0 && age < win && age >= mine);
}
__device__ __forceinline__ void insert16_retry(u32* w, u32 old, u32 sh, u32 val, u32 mine, u32 ip, u32 win) {
    while (insert16_wants(old, sh, mine, ip, win)) {
        const u32 prev = atomicCAS(w, old, (old & ~(0xFFFFu << sh)) | (val << sh));
        if (prev == old) break;
        old = prev;
    }
}
// Ordered insert: the four waves of the workgroup write their 16-bit entries one wave after the
// other (plain stores, a barrier between the waves), so a later wave -- later positions -- overwrites
// an earlier one.  Two lanes of ONE wave that hit the same bucket in the same store instruction are
// arbitrated by the LDS hardware, a fixed function of the lane numbers: whichever it picks, the table
// content is the same on every run (what the format needs is determinism; "most recent wins" is only
// a preference).  Costs four barriers instead of a compare-and-swap per entry with retries, which
// measured 1400-1900 cycles per window (stride 1) and 2800-4000 (stride 2).
#ifndef ZL_INSERT_CAS
#define ZL_INSERT_CAS 0
#endif
__device__ __forceinline__ void insert16_pair(u32* tl, u32 bl, u32 old_l, u32* ts, u32 bs, u32 old_s, u32 p, u32 ip, u32 win) {
    const u32 val = (p + 1) & 0xFFFF;
    if (val == 0) return;
    u32* wl = tl + (bl >> 1);
    u32* ws = ts + (bs >> 1);
    const u32 shl = (bl & 1) * 16, shs = (bs & 1) * 16, mine = p - ip;
    const bool go_l = insert16_wants(old_l, shl, mine, ip, win), go_s = insert16_wants(old_s, shs, mine, ip, win);
    u32 prev_l = old_l, prev_s = old_s;
    if (go_l) prev_l = atomicCAS(wl, old_l, (old_l & ~(0xFFFFu << shl)) | (val << shl));
    if (go_s) prev_s = atomicCAS(ws, old_s, (old_s & ~(0xFFFFu << shs)) | (val << shs));
    if (go_l && prev_l != old_l) insert16_retry(wl, prev_l, shl, val, mine, ip, win);
    if (go_s && prev_s != old_s) insert16_retry(ws, prev_s, shs, val, mine, ip, win);
}

// ---------------------------------------------------------------------------------------
// E1: match finding.  chunk c covers src[c*chunk_size, +len); output per chunk:
//   lits[c*lit_stride ...]   literal bytes in order
//   seqs[c*seq_stride ...]   {lit_cum, match_len, offset, pos} per sequence
//   meta[c]                  {nseq, nlit}
// Window by window the kernel runs one of two bodies.  Stride 1: every position is probed, 256 bytes
// per iteration.  Stride 2: every iteration covers 512 bytes; each lane PROBES one even position and
// INSERTS two (its even position and the odd one behind it), so the tables know every position and a
// repeat is found wherever it started -- at most one byte late -- while the per-position work that
// dominates stride 1 (hashing aside: three candidate gathers, verification, the parse links) is done
// for half the positions (cf. the step > 1 of ZSTD_compressBlock_fast, zstd.c:29340, which also probes
// a subset and inserts around matches).  Its parse runs over the 256 even "nodes" of the window; an
// odd position is a literal by construction: a lane without a match emits its two bytes, a lane whose
// match ends on an odd position emits the byte behind it.
// Stride 1 is used while the data is dense in matches, stride 2 otherwise.  Measured with each body
// alone: on text-like input stride 2 costs 2.7 % of ratio (matches closer than one window are
// invisible to the tables, and the window is twice as long) for 27 % less match time; on inputs with
// few matches it costs nothing (iid10 even gains 0.8 %) and saves 25-40 %.  The switch is a pure
// function of the data (sequences found in the previous window), so the output stays deterministic.
// ---------------------------------------------------------------------------------------
#define WIN2 (2 * ENC_THREADS)
// Round 6: the bulk instantiation at EIGHT workgroups per CU -- 2^12 + 2^12 entries (18.4 KiB of LDS with the rest; 2^13 + 2^11 was 22.4: seven) and 64
// vector registers (65 before: __launch_bounds__' second argument asks for them).  Not for the latency it hides (text alone: 9.15 -> 8.90 ms per GiB)
// but for what a launch is made of: the chunks of a kind take equally long, a launch whose expensive kind has more chunks than the device has
// places runs that kind in two rounds, and the second round is a handful of workgroups on an empty device -- the 1 GiB mix has 2048 chunks of text
// for 1792 places: 3.33 ms where its four kinds add up to 3.0.  With 2048 places: 2.93 ms.  The long table's half costs text 2.9 % of ratio, the short
// table's double gives 1.8 back (4.257 -> 4.210; the mix 1.634 -> 1.633); 2^12 + 2^11 (ten places by LDS, eight by registers): 2.88 ms, text 4.133.
#ifndef LONG_LOG_BULK
#define LONG_LOG_BULK 12
#endif
// ... and the one-frame (windowed) instantiation likewise: 2^12 + 2^12 entries and 64 vector registers (two spilled) = eight workgroups per CU where 2^13 + 2^12 and 76
// registers gave six.  1 GiB mix in one frame: match 4.84 -> 4.17 ms, 64 MiB of text 0.997 -> 1.021 of the reference's size, the python sources (three copies:
// above the deep finder's 16 MiB) 1.017 -> 1.028, records unchanged.  (Round 5 tried SEVEN through the short table alone: 4.88 -> 4.75 ms, not taken.)
#ifndef LONG_LOG_FAR
#define LONG_LOG_FAR 12
#endif
#ifndef ZL_FAR_MIN_BLOCKS
#define ZL_FAR_MIN_BLOCKS 8
#endif
#ifndef ZL_BULK_MIN_BLOCKS
#define ZL_BULK_MIN_BLOCKS 8    // __launch_bounds__' second argument for the bulk instantiation: workgroups per CU the register allocation must allow
#endif
#ifndef ZL_MATCH_SGPRS
#define ZL_MATCH_SGPRS 88  // a CU admits seven 256-thread workgroups up to 96 scalar registers per wave (six from 97; MI355X_MICROARCH.md, Residency); the lean body took the bulk instantiation to 97
#endif
#ifndef ZL_SPARSE_MIN
#define ZL_SPARSE_MIN 5
#endif
#ifndef ZL_CHEAP_FAR_MIN
#define ZL_CHEAP_FAR_MIN 12  // chunks of cheap bytes take a far hint only when the far probe measured at least this many bytes
#endif
#ifndef ZL_CHEAP_BITS
#define ZL_CHEAP_BITS 36  // tenths of a bit: chunks whose bytes cost less keep only long matches (0: off)
#endif
#ifndef ZL_WBUF
#define ZL_WBUF 1
#endif
#ifndef ZL_LEAN_DEDUPE
#define ZL_LEAN_DEDUPE 0  // (measured: 8.45 -> 8.67 ms per 896 MiB of text with it)
#endif
#ifndef ZL_LEAN
#define ZL_LEAN 1  // the bulk path's interior windows take the lean stride-1 body (0: the general body everywhere)
#endif
#ifndef ZL_DENSE_SEQS
#define ZL_DENSE_SEQS 24  // sequences per 512 bytes from which on the stride-1 body is used
#endif
#ifndef ZL_DENSE_SEQS_FAR
#define ZL_DENSE_SEQS_FAR 8  // the same for the windowed layout, whose callers sit behind PCIe and asked for the reference's ratio
#endif
#ifndef ZL_SUB8_DIV
#define ZL_SUB8_DIV 8  // one frame: the eight-lane look-up-and-insert phases while at least 1 in 8 of a window's sequences has an offset below 64
                       // (1 in 4: match 7.52 ms, python sources 4.373; 1 in 8: 7.55 / 4.390; 1 in 16: 7.82 / 4.403; always, round 3: 8.00 / 4.420)
#endif
#ifndef ZL_SUB_LANES
#define ZL_SUB_LANES 8   // dense body, look-up-and-insert phases: lanes of a wave that read and insert together (64: the whole wave at once)
#endif
#ifndef ZL_DENSE_LIT_PCT_FAR
#define ZL_DENSE_LIT_PCT_FAR 65  // windowed layout: 0 = by sequence count (above); else the dense body while literals are at most this percentage of a window
#endif
// DICT: a dictionary's content sits in front of every chunk.  A template parameter: without it the
// position arithmetic, the pointer selects and the straddle tests of the dictionary fold away (5.93 -> 5.80 ms).
// PROF: the in-kernel phase stamps of ZL_c_profile = 2 (compiled out otherwise: 0.7 %).
// FAR: the windowed (single-frame) layout.  The chunk is one run of blocks of a larger frame, so offsets may reach in
// front of it: every 32-byte neighbourhood brings at most one offset found by the far table (zl_far_*_kernel below),
// tried as a fourth candidate by every lane of the neighbourhood -- the lane the table hit extends it forwards, the
// lanes in front of it backwards -- and a repeat offset may point in front of the chunk as well.
// INTRA: the intra-window table (see the dense body) is compiled in; it is used by every chunk of the windowed layout and,
// elsewhere, by the chunks whose descriptor asks for it: inputs of at most one chunk, where most of what repeats does so
// within a few hundred bytes (a 1000-byte serialized data frame: 585 -> about 470 bytes).  The bulk path of large inputs
// in the independent-frames layout runs without it (three more barriers per window).
// A short match at a long distance costs more than the literals it replaces (an offset of 2^k takes k extra bits on top
// of the three codes; a literal of compressible data 3-5 bits): matches below ZL_SHORT_LEN bytes are kept only with a
// repeat offset or an offset below 2^ZL_SHORT_OFFLOG.
#ifndef ZL_SHORT_LEN
#define ZL_SHORT_LEN 0
#endif
#ifndef ZL_SHORT_OFFLOG
#define ZL_SHORT_OFFLOG 13
#endif
// (one-frame layout, whose short hash covers 5 bytes: a 5-byte match needs an offset below 8 KiB.  64 MiB of iid10
// 2.236 -> 2.276, python sources 4.321 -> 4.316, text unchanged; 7 bytes / 4 KiB: iid10 2.305 but the sources 4.293.
// The independent-frames layout runs without the rule: its 6-byte hash finds few such matches, and there the rule cost
// the sources 0.8 % for nothing.)
#ifndef ZL_SHORT_LEN_FAR
#define ZL_SHORT_LEN_FAR 6
#endif
// The short table's candidates are fetched only from this far back (0 = as far as the table knows).
#ifndef ZL_SHORT_REACH
#define ZL_SHORT_REACH 0
#endif
template <bool FAR>
__device__ __forceinline__ bool short_in_reach(u32 dist) { return FAR || ZL_SHORT_REACH == 0 || dist <= ZL_SHORT_REACH; }
#ifndef ZL_BACK_BULK
#define ZL_BACK_BULK 0
#endif
#ifndef ZL_BACK_MAX
#define ZL_BACK_MAX 8   // bytes a sequence of the stride-1 body may reach back into the literals in front of it (0: off)
#endif
template <bool FAR>
__device__ __forceinline__ bool worth_it(u32 len, u32 off, u32 rep) {
    const u32 need = FAR ? ZL_SHORT_LEN_FAR : ZL_SHORT_LEN;
    return need == 0 || len >= need || off == rep || off < (1u << ZL_SHORT_OFFLOG);
}
// HI: the operating point of levels >= 5 (no dictionary): tables of 2^14 + 2^13 entries (48 KiB of LDS: three workgroups per
// CU instead of seven) and verification up to 72 bytes per lane.  The CPU model of the parse (scripts/dev/sim/sim.c) gives
// text +3.2 %, python sources +5.3 % of ratio for it; what the device gives and costs is in DESIGN.md 3.3.
template <bool DICT, bool PROF, bool FAR, bool INTRA, bool HI = false>
__global__ __launch_bounds__(ENC_THREADS, (!DICT && !PROF && !FAR && !INTRA && !HI) ? ZL_BULK_MIN_BLOCKS : ((!DICT && !PROF && FAR && !HI) ? ZL_FAR_MIN_BLOCKS : 1)) __attribute__((amdgpu_num_sgpr(ZL_MATCH_SGPRS))) void zl_match_auto_kernel(zl_match_args a, u32 nchunks) {
    static_assert(!(DICT && HI), "the dictionary's table image has the base sizes");
    constexpr int LLOG = HI ? LONG_LOG_HI : ((!DICT && !FAR && !INTRA) ? LONG_LOG_BULK : (FAR ? LONG_LOG_FAR : LONG_LOG));
    constexpr u32 lane_cap = (FAR || HI) ? LANE_MATCH_CAP_FAR : LANE_MATCH_CAP;
    // The bulk path (no dictionary, independent frames, inputs larger than a chunk) runs with half the short table: its
    // candidates are nearer (mean age = table size), so their gathers hit closer caches -- 1 GiB mix: match 5.74 -> 4.98 ms
    // for 0.1 % of ratio (text -1.2 %, python sources -2.4 %; a distance cap on the short candidates bought the same time
    // for twice the ratio: DESIGN.md appendix A.1).  The other instantiations keep 2^12 entries.
    // One frame at level 3: 2^13 + 2^13 entries (34 KiB of LDS: four workgroups per CU).  Round 2 ran this layout with the level >= 5
    // tables (2^14 + 2^13: three per CU) at every level; measured on the 1 GiB mix: match 10.3 -> 7.9 ms, python sources 4.396 ->
    // 4.368 (1.014x the reference's size), text 4.604 -> 4.533 (reference 4.512).  2^14 + 2^11 ran as fast and cost the sources
    // twice as much.
    constexpr int SLOG = HI ? SHORT_LOG_HI : (FAR ? SHORT_LOG_FAR : ((DICT || INTRA) ? SHORT_LOG : SHORT_LOG_BULK));
    constexpr bool WB = ZL_WBUF && !DICT;  // the stride-1 bodies read the own bytes of a window from S.wbuf (the positions of a dictionary frame are virtual: not built)
    __shared__ MatchSharedT<LLOG, SLOG, WB> S;
    ZL_POISON(S, 4);
    const u32 t = threadIdx.x;
    const u32 c = a.order ? a.order[zl_spread(blockIdx.x, nchunks)] : zl_spread(blockIdx.x, nchunks);
#if ZL_MATCH_PRIO
    // (schedule only) Longest first leaves the expensive chunks that did not fit into the launch's first round -- 2048 chunks of text on 1792
    // resident workgroups: 256 of them -- running beside cheap ones, and the launch lasts as long as they do.  Their waves go first where a
    // SIMD has to choose: the chunk's class sits behind the order (zl_launch_chunk_order: cost_scratch = order + nchunks).
    if (a.order && a.order[nchunks + c] >= 3u) __builtin_amdgcn_s_setprio(ZL_MATCH_PRIO);
#endif
    const zl_chunk_desc cd = a.chunks[c];
    const u8* src = a.src + cd.src_off;
    const u32 n = cd.src_size;
    u8* lits = a.lits + (size_t)c * a.lit_stride;
    zl_seq* seqs = a.seqs + (size_t)c * a.seq_stride;

    // With a dictionary, its last D content bytes sit in front of the chunk as history: positions are
    // counted from the start of that content (chunk position p is "virtual" position D + p), the match
    // tables start out as they look after the dictionary's last byte, and a candidate below D is read
    // from the dictionary (zstd.c:30120, the dictionary variant of the double-hash parse).
    const u32 D = DICT ? a.dict_len : 0u;
    const u8* const dict = a.dict;
    static_assert(!(DICT && FAR), "a dictionary frame is a frame of its own");
#ifdef ZL_TRACE
    constexpr bool LEAN = ZL_LEAN && !DICT && !FAR && !INTRA && !HI;  // (the stamped instantiation is the profiling one)
    u32 tr_it = 0;
    const bool tr_on = PROF && a.stats != nullptr && c == (u32)(ZL_TRACE) && (threadIdx.x & 63) == 0;
#define TR(k_) do { if (tr_on && tr_it < ZL_TRACE_ITERS) { asm volatile("" ::: "memory"); a.stats[16 + (tr_it * 4 + (threadIdx.x >> 6)) * ZL_TRACE_POINTS + (k_)] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } } while (0)
#else
    constexpr bool LEAN = ZL_LEAN && !DICT && !PROF && !FAR && !INTRA && !HI;
#define TR(k_) do { } while (0)
#endif
    // bytes of the frame in front of this chunk INSIDE THE CHUNK'S SECTION: a source position is a chunk position that may be
    // "negative", down to -reach_sec.  (Until round 3 the repeat-offset, far-hint and one-byte backward-extension checks used the
    // bytes in front of the chunk in the whole FRAME: the stride-2 body's backward extension could read -- and make a sequence
    // copy -- the byte in front of a section's first byte, which a pledged stream's later jobs do not even have on the device:
    // a memory fault that scripts/fuzz_stream.py found.)
    const u32 reach_sec = FAR ? (u32)(cd.src_off & (((u64)ZL_CHUNK << ZL_FAR_SECTION_LOG) - 1)) : 0u;
    const u32* const hints = FAR ? a.far_hints + (size_t)c * ZL_FAR_HINT_WORDS : nullptr;
    auto at = [&](u32 q) -> const u8* {  // where (virtual) position q lives
        if (FAR) return src + (ptrdiff_t)(i32)q;
        return q >= D ? src + (q - D) : dict + q;
    };
    for (u32 i = t; i < (1u << LLOG) / 2; i += ENC_THREADS) S.tbl_long[i] = D ? a.dict_hash[i] : 0u;
    for (u32 i = t; i < (1u << SLOG) / 2; i += ENC_THREADS) S.tbl_short[i] = D ? a.dict_hash[(1u << LONG_LOG) / 2 + i] : 0u;
    if (t == 0) { S.rep = 0; S.cand_end = 0; }
    const bool intra_on = INTRA && (FAR || (cd.flags & 1u));  // uniform
    // (a launch that mixes small inputs with chunks of large ones -- a batch -- must write for every chunk what a launch of
    // its own would: the chunks of large inputs hash as the bulk instantiation does)
    const u32 sshift = (INTRA && !DICT && !FAR && !HI && !intra_on) ? 64u - SHORT_LOG_BULK : 64u - (u32)SLOG;
    const u32 lshift = (INTRA && !DICT && !FAR && !HI && !intra_on) ? 64u - LONG_LOG_BULK : 64u - (u32)LLOG;  // (likewise the long table, should the bulk path's ever be the smaller)
    // Inputs of at most 16 KiB (no dictionary): the short table hashes FOUR bytes and a table candidate of four bytes is a match, as
    // the reference's level 3 does for such inputs (zstd.c:29607: minMatch 4 up to 16 KiB); other single-chunk inputs hash five
    // (zstd.c:29581).  A pickled data frame of 1 KiB held 9 sequences against the reference's 56 (1.29x its size): its repeats are
    // 4 and 5 bytes long, 12 bytes apart.  Uniform; folds to the constants of the layout where the instantiation knows them.
    const bool tiny = INTRA && !DICT && !FAR && !HI && intra_on && n <= 16384u;
    const u32 sbsh = FAR ? 64u - 8u * ZL_SHORT_BYTES_FAR : (tiny ? 32u : ((INTRA && !DICT && !HI && intra_on) ? 24u : 64u - 8u * ZL_SHORT_BYTES));
    const u32 tmin0 = tiny ? MIN_MATCH : MIN_MATCH + 1;  // shortest match a table candidate may be (a repeat offset: MIN_MATCH)
    // backward extension of matches (stride-1 body): everywhere but on the bulk path -- the chunks of large inputs at level 3
    // in the independent-frames layout, whose measure is GB/s: there it cost 6.7 % of match time (5.09 -> 5.43 ms) for 0.4 %
    // of the mix's ratio (text +4.0 %, python sources +1.2 %).  Uniform.
    const bool back_on = ZL_BACK_MAX != 0 && (ZL_BACK_BULK || FAR || HI || DICT || intra_on);
    lds_barrier();
    // ---- What is a match worth?  A sequence with a new offset costs some 24 bits (an offset of 2^k takes k extra bits on top of
    // the three codes); the literals it replaces cost what the chunk's bytes cost.  Where bytes are CHEAP -- the order-0 entropy
    // of 16 KiB sampled from the chunk's four quarters is below ZL_CHEAP_BITS/10 bits, e.g. i.i.d. data over a dozen symbols -- a new-offset match of
    // fewer than eight bytes loses bits, and looking for such matches is all that the search does there: i.i.d. bytes over ten
    // letters took 1.47 ms per 256 MiB for 0.5 sequences per 512 bytes that made the output LARGER than their literals would
    // have (2.344 with them, 2.35 without).  Such chunks keep only matches of eight bytes and more (or a repeat offset), their
    // windows are then mostly match-free, and the search acceleration below skips seven windows in eight, as on random bytes.
    // Level 3 and below, chunks of large inputs in either layout (the level >= 5 instantiations, dictionaries and small inputs keep
    // the fixed minimum: their ratios are pinned); a pure function of the chunk's bytes.
    u32 smin = ZL_SPARSE_MIN;  // stride-2 body: shortest new-offset match kept (uniform)
    bool cheap = false, noisy = false;
    bool dull = false;  // cheap or noisy bytes AND hardly a match in the chunk's first 4 KiB: the search acceleration below ramps up twice as fast and twice as far (uniform)
    if ((LEAN || (FAR && !HI)) && ZL_CHEAP_BITS && n >= 16384) {
        u32* const bins = S.cand;  // [256]
        bins[t] = 0;
        lds_barrier();
        for (u32 part = 0; part < 4; part++) {  // 4 KiB at the start of every quarter of the chunk
            const uint4 x = ld128(src + (size_t)part * (n >> 2) + 16 * t);
            const u32 w4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int k = 0; k < 4; k++) { atomicAdd(&bins[w4[k] & 0xFF], 1u); atomicAdd(&bins[(w4[k] >> 8) & 0xFF], 1u); atomicAdd(&bins[(w4[k] >> 16) & 0xFF], 1u); atomicAdd(&bins[w4[k] >> 24], 1u); }
        }
        lds_barrier();
        const u32 c = bins[t];
        float bits = c ? (float)c * (14.0f - __log2f((float)c)) : 0.0f;  // c * log2(16384 / c)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) bits += __shfl_xor(bits, d, 64);
        if ((t & 63) == 0) S.scan_b[t >> 6] = __float_as_uint(bits);
        lds_barrier();
        const float total = __uint_as_float(S.scan_b[0]) + __uint_as_float(S.scan_b[1]) + __uint_as_float(S.scan_b[2]) + __uint_as_float(S.scan_b[3]);
        cheap = total * 10.0f < (float)ZL_CHEAP_BITS * 16384.0f;
        noisy = total * 10.0f >= 78.0f * 16384.0f;  // 7.8 bits and more per byte: random for all an order-0 look can tell
        lds_barrier();
    }
    // (cheap bytes alone do not say it: records that repeat, a column of doubles are cheap byte for byte AND all matches, short
    // ones between the long ones included.  The chunk's first 4 KiB are searched as ever; the rule starts behind them where
    // matches covered less than a tenth of those bytes.)
    u32 probed = 0, probed_lits = 0;
#define tmin (smin > tmin0 ? smin : tmin0)  /* stride-1 bodies: shortest match a table candidate may be */

    // ---- literals are not copied while a chunk has no sequence.  Random bytes, i.i.d. bytes over a few letters -- half of the bench
    // mix -- yield none: their windows are emitted "as literals" seven times in eight without a search (below), and all that did was
    // copy the chunk, 16 bytes per lane, into a buffer from which the entropy stage read it again (the kernel wrote
    // 1.67 GB of intermediates per GiB).  While nseq == 0 every byte in front of the current window is a literal and nlit == ip: the
    // windows without a match store nothing, the entropy kernel reads such a chunk's literals from the source itself
    // (meta.nseq == 0), and the first window that does produce a sequence brings the bytes in front of it in, once.
#define ZL_LITS_CATCH_UP(nlit_before_) do { if (nseq == 0 && tot_s != 0 && (nlit_before_) != 0) { const u32 cnt__ = (nlit_before_), body__ = cnt__ & ~15u; \
        for (u32 q__ = t * 16; q__ < body__; q__ += ENC_THREADS * 16) { uint4 x__; __builtin_memcpy(&x__, src + q__, 16); __builtin_memcpy(lits + q__, &x__, 16); } \
        if (t < cnt__ - body__) lits[body__ + t] = src[body__ + t]; } } while (0)
    u32 nseq = 0, nlit = 0;  // uniform running totals
    u32 miss = 0;            // consecutive windows without a sequence (uniform)
    u32 carry = 0;           // where the parse enters the next window, in bytes (uniform)
    bool dense = !a.sparse_only || D != 0;  // which body the next window gets (uniform); dictionary frames use the stride-1 body throughout
    u32 wbuf_ip = 0xFFFFFFFFu;  // lean body: the position S.wbuf starts at (uniform)
    bool sub8 = true;           // stride-1 body with INTRA: look up and insert eight lanes after eight lanes (uniform; see there)
    u64 v_pref = 0;          // stride 1: prefetched bytes of position pref_ip + t
    u32 pref_ip = 0xFFFFFFFFu;
    u64 pref0 = 0, pref1 = 0;  // stride 2: prefetched bytes [p-1, p+15) of position pref_ip2 + 2t
    u32 pref_ip2 = 0xFFFFFFFFu;
    u32 hint_pref = 0;       // FAR: the hint word of the prefetched position
    u32 hint_ip = 0xFFFFFFFFu;  // stride 1: the window hint_pref was fetched for
    const bool prof = PROF && a.stats != nullptr && t == 0;
    u64 c_find = 0, c_parse = 0, c_emit = 0, c_ins = 0, c_it = 0;
    for (u32 ip = 0; ip < n;) {
        u32 win, win_seqs = 0, win_lits = 0;
        // ---- the stride-1 body for the windows of the bulk path that lie well inside the chunk (both this window's and the next
        // one's reads are in bounds, the window's own bytes were prefetched): the same candidates, the same parse, the same
        // bytes written as the general body below, but no bounds tests, no per-lane conditions held as wave masks (the
        // compiler keeps a per-lane bool in a pair of scalar registers and combines them on the CU's ONE scalar unit, which
        // the 28 resident waves share: 184 scalar instructions per wave and window in round 3, most of them mask logic and
        // exec-mask saves around one-line conditionals), selects instead of branches, candidates fetched by every lane (a lane
        // without one reads its own position: the same line its neighbours read).
        // ---- stride-1 windows well inside the chunk read their own bytes -- the eight of every lane, and the words the extension
        // loop compares -- from S.wbuf: 22 lanes fetch 16 bytes each where 256 lanes fetched eight overlapping bytes each and every
        // extension round 64 more.  The bytes of the next window are asked for at the top and put in place behind the barrier
        // that ends the reads this window makes of the buffer.
        const bool wb_ok = WB && dense && ip + 2 * WIN + 8 <= n;       // (uniform; the reads of this window and of the next are in bounds)
        const bool wb_next = wb_ok && ip + 3 * WIN + 8 <= n;           // (the next window qualifies as far as its bounds go)
        uint4 wb_pref = make_uint4(0, 0, 0, 0);
        if (wb_ok) {
            if (wbuf_ip != ip) {  // (the window in front took another body: chunk start, a stride-2 stretch)
                if (t < ZL_WBUF_BYTES / 16) *(uint4*)(S.wbuf + 16 * t) = ld128(src + ip + 16 * t);
                lds_barrier();
            }
            if (wb_next && t < ZL_WBUF_BYTES / 16) wb_pref = ld128(src + ip + WIN + 16 * t);
        }
        // eight bytes at window offset o: three aligned words, funnel-shifted
        auto wb64 = [&](u32 o) -> u64 {
            const u32* const w = (const u32*)S.wbuf + (o >> 2);
            const u32 d0 = w[0], d1 = w[1], d2 = w[2];
            const u32 sh = o & 3;
            return (u64)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
        };
        // sixteen bytes at window offset o against sixteen bytes in memory: how many agree from the start
        auto wb_common16 = [&](u32 o, const u8* theirs) -> u32 {
            const uint4 b = ld128(theirs);
            const u32* const w = (const u32*)S.wbuf + (o >> 2);
            const u32 d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
            const u32 sh = o & 3;
            const u64 a_lo = (u64)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
            const u64 a_hi = (u64)__builtin_amdgcn_alignbyte(d3, d2, sh) | ((u64)__builtin_amdgcn_alignbyte(d4, d3, sh) << 32);
            const u32 l0 = common_bytes(a_lo, (u64)b.x | ((u64)b.y << 32)), l1 = common_bytes(a_hi, (u64)b.z | ((u64)b.w << 32));
            return l0 + (l0 == 8 ? l1 : 0u);
        };
        auto wb_advance = [&]() {  // behind the barrier that ends the reads of the buffer
            if (wb_next && t < ZL_WBUF_BYTES / 16) *(uint4*)(S.wbuf + 16 * t) = wb_pref;
            wbuf_ip = wb_next ? ip + WIN : 0xFFFFFFFFu;
        };
        if (LEAN && wb_ok) {
            win = WIN;
            const u32 p = ip + t;
            const u64 v = wb64(t);
            TR(0);
            const u32 rep = S.rep;
            const u32 hl = hash_long(v, lshift), hs = hash_short_rt(v, sbsh, sshift);
            u32 best_len = 0, best_off = 0;
            if (carry < WIN) {  // (otherwise the whole window lies inside an earlier match: nothing to search)
                const u32 el = ((const unsigned short*)S.tbl_long)[hl], es = ((const unsigned short*)S.tbl_short)[hs];
                // distance to the nearest position an entry can mean; 0 = no candidate (empty entry, the position itself 64 KiB on, in front of the chunk)
                u32 dl = el ? ((p + 1 - el) & 0xFFFF) : 0u, ds = es ? ((p + 1 - es) & 0xFFFF) : 0u;
                dl = dl - 1 < p ? dl : 0u;
                ds = ds - 1 < p ? ds : 0u;
                const u32 dr = rep - 1 < p ? rep : 0u;
#ifdef ZL_TRACE
                if (tr_on && (dl + ds + dr) == 0xFFFFFFF0u) tr_it += 0;  // (makes the stamp wait for the look-ups)
#endif
                TR(1);
#if ZL_LEAN_DEDUPE
                // a lane-load is an L1 tag look-up: only lanes that have a candidate fetch it, and a position named twice is fetched once
                u64 w_r = 0, w_l = 0, w_s = 0;
                if (dr) w_r = ld64(src + (p - dr));
                if (dl && dl != dr) w_l = ld64(src + (p - dl));
                if (ds && ds != dr && ds != dl) w_s = ld64(src + (p - ds));
                w_l = dl == dr ? w_r : w_l;
                w_s = ds == dr ? w_r : (ds == dl ? w_l : w_s);
#else
                const u64 w_r = ld64(src + (p - dr)), w_l = ld64(src + (p - dl)), w_s = ld64(src + (p - ds));
#endif
                u32 l_r = common_bytes(v, w_r), l_l = common_bytes(v, w_l), l_s = common_bytes(v, w_s);
                l_r = dr ? l_r : 0u; l_l = dl ? l_l : 0u; l_s = ds ? l_s : 0u;
                // repeat offset from 4 bytes, table candidates from 5; the long table's wins a tie against the short one's
                best_len = l_r >= MIN_MATCH ? l_r : 0u;
                best_off = dr;
                const u32 floor_t = smin > MIN_MATCH + 1 ? smin - 1 : MIN_MATCH;  // (a table candidate must be longer than this)
                const bool take_l = l_l > (best_len > floor_t ? best_len : floor_t);
                best_len = take_l ? l_l : best_len; best_off = take_l ? dl : best_off;
                const bool take_s = l_s > (best_len > floor_t ? best_len : floor_t);
                best_len = take_s ? l_s : best_len; best_off = take_s ? ds : best_off;
                TR(2);
                // extend while whole 8-byte words keep matching.  (Fetching the 32 bytes behind a full match on both sides AT ONCE --
                // one round trip instead of up to four dependent ones, which are 3800 of an iteration's 11 000 cycles -- was measured
                // SLOWER, 9.3 -> 11.9 ms per 896 MiB of text: every lane-load is a tag look-up in the CU's L1, the L1 makes one per
                // clock, and it is the match kernel's bound; the loop's later rounds have few lanes left.  profiles/r04_match_timeline*.)
                // (sixteen bytes per round: half the dependent round trips; 8.42 -> 7.98 ms per 896 MiB of text.  All 32 bytes at once: 8.67)
                static_assert((lane_cap - 8) % 16 == 0, "8 + 16 k");
                while (best_len >= 8 && (best_len & 7) == 0 && best_len < lane_cap) {
                    const u32 l = wb_common16(t + best_len, src + (p - best_off) + best_len);
                    best_len += l;
                    if (l < 16) break;
                }
            }
            TR(3);
            bool has_match = best_len >= MIN_MATCH;
            S.cand[t] = has_match ? (best_len | (best_off << 8)) : 0u;
            lds_barrier();
            wb_advance();
            TR(4);
            // one-step lazy evaluation against the neighbour's candidate (the window's last lane reads cand_end = 0: a gain of 0, below any match's)
            {
                const u32 nx = S.cand[t + 1];
                const int g1 = 4 * (int)best_len - (int)(31 - __clz(best_off + 1)) + 4;
                const int g2 = 4 * (int)(nx & 0xFF) - (int)(31 - __clz((nx >> 8) + 1));
                has_match = has_match && g2 <= g1;
            }
            const u32 wbase = t & ~63u, wave = t >> 6;
            u32 jr[6];
            u32 jp = t + (has_match ? best_len : 1u);
    #pragma unroll
            for (int r = 0; r < 6; r++) {
                jr[r] = jp;
                const u32 far = __shfl(jp, (int)((jp - wbase) & 63), 64);
                jp = (jp < wbase + 64) ? far : jp;
            }
            S.jump[0][t] = (u16)jp;
            TR(5);
            lds_barrier();
            TR(6);
            u32 enter = carry, my_entry = 0xFFFFFFFFu;
    #pragma unroll
            for (u32 k = 0; k < 4; k++) {
                if (enter < 64 * (k + 1)) {
                    if (k == wave) my_entry = enter;
                    enter = S.jump[0][enter];
                }
            }
            carry = enter - WIN;
            TR(7);
            const bool on_path = zl_path_marks(jr, t, wbase, my_entry);
            TR(8);
            const bool is_seq = on_path && has_match, is_lit = on_path && !has_match;
            const u64 ms = __ballot(is_seq), ml = __ballot(is_lit);
            const u64 below = (1ull << (t & 63)) - 1;
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
            TR(9);
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
            TR(10);
            for (u32 ph = 0; ph < 4; ph++) {
                if (ph == wave) {
                    const unsigned short val = (unsigned short)(p + 1);
                    ((unsigned short*)S.tbl_long)[hl] = val;
                    ((unsigned short*)S.tbl_short)[hs] = val;
                }
                if (ph < 3) lds_barrier();
            }
            ZL_LITS_CATCH_UP(nlit);
            nseq += tot_s;
            nlit += tot_l;
            win_seqs = tot_s;
            win_lits = tot_l;
            TR(11);
            lds_barrier();
            TR(12);
#ifdef ZL_TRACE
            tr_it++;
#endif
        } else if (dense) {
            win = WIN;
            const u64 s0 = prof ? __builtin_amdgcn_s_memtime() : 0;
            const u32 rep = S.rep;
            u64 v[PPT];
            u32 far_off[PPT];
            u32 hl[PPT], hs[PPT], best_len[PPT], best_off[PPT], q_best[PPT];
            u32 word_l[PPT], word_s[PPT];  // the table words as read by the lookup (reused by the insert)
            bool in_range[PPT], can_match[PPT];
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                const u32 p = ip + t + ENC_THREADS * j;
                in_range[j] = p < n;
                can_match[j] = p + 8 <= n;  // the last 7 bytes of a chunk are always literals
                v[j] = 0;
                if (wb_ok) v[j] = wb64(t);
                else if (pref_ip == ip) v[j] = v_pref;  // fetched while the previous window was being searched
                else if (can_match[j]) v[j] = ld64(src + p);
                else if (in_range[j]) v[j] = src[p];
                if (FAR) {
                    const u32 hw = (hint_ip == ip) ? hint_pref : (in_range[j] ? hints[p >> ZL_FAR_HINT_LOG] : 0u);
                    far_off[j] = (smin > ZL_SPARSE_MIN && (hw >> 26) < ZL_CHEAP_FAR_MIN) ? 0u : (hw & 0x3FFFFFFu);  // (cheap bytes: a far offset costs 30 bits)
                }
                hl[j] = hs[j] = best_len[j] = best_off[j] = q_best[j] = 0;
                word_l[j] = word_s[j] = 0;
            }
            const bool search = carry < WIN;  // otherwise the whole window lies inside an earlier match
            TR(0);
            // candidates of both positions are fetched together: one memory round trip.  A lane without
            // a candidate reads its own position and is masked out afterwards.
            u64 w_r[PPT], w_l[PPT], w_s[PPT], w_f[PPT];
            u32 q_r[PPT], q_l[PPT], q_s[PPT];
            bool have_r[PPT], have_l[PPT], have_s[PPT], have_f[PPT];
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                const u32 p = ip + t + ENC_THREADS * j;
                have_r[j] = have_l[j] = have_s[j] = have_f[j] = false;
                w_f[j] = 0;
                const u32 V = D + p;  // virtual position
                q_r[j] = q_l[j] = q_s[j] = V;
                u32 el = 0, es = 0;
                if (can_match[j]) {
                    hl[j] = hash_long(v[j], lshift);
                    hs[j] = hash_short_rt(v[j], sbsh, sshift);
                }
                if (INTRA && intra_on) {
                    // LOOK UP AND INSERT, wave after wave: wave k reads its entries and puts its own positions in their place before
                    // wave k + 1 reads, so a lookup sees every position up to the start of its own wave's 64 -- the window's own
                    // included (the next line of a listing, the next record) -- and the tables end up exactly as the insert phases
                    // at the end of the window left them (the last wave's positions win).  Round 2 kept the window's own positions
                    // in a third table of 256 slots (3 barriers + a gather of its own) AND ran the insert phases (3 more).
                    unsigned short* const tl = (unsigned short*)S.tbl_long;
                    unsigned short* const ts = (unsigned short*)S.tbl_short;
                    for (u32 ph = 0; ph < 4; ph++) {
                        if (ph == (t >> 6)) {
                            // ... and inside a wave, eight lanes after eight lanes (a wave's LDS accesses execute in program order: no
                            // barrier needed): a lookup sees everything up to 8 positions in front of it.  What that buys is the
                            // NEAREST repeat of records of 8 or 16 bytes -- a column of doubles, R's numeric vector -- whose
                            // neighbours differ in one or two bytes: seq(0, by = 0.25) matched its ninth-last value and paid a full byte
                            // per literal where the last value leaves 2 to 4 bits (1.57x the reference's size; tests/ratio_inputs.py).
                            // The eight sub-phases are 32 dependent LDS operations per wave, the waves one behind the other: 3900 of an
                            // iteration's 13 900 cycles on text in one frame (scripts/dev/trace_match.py), where they buy nothing.  Chunks of one
                            // frame take them only while short offsets are what the data yields (sub8, decided window by window below).
                            if (sub8) {
#pragma unroll
                            for (u32 sub = 0; sub < 64 / ZL_SUB_LANES; sub++) {
                                if (((t & 63) / ZL_SUB_LANES) == sub && can_match[j]) {
                                    el = tl[hl[j]]; es = ts[hs[j]];
                                    const unsigned short val = (unsigned short)(V + 1);
                                    tl[hl[j]] = val; ts[hs[j]] = val;
                                }
                                asm volatile("" ::: "memory");
                            }
                            } else if (can_match[j]) {  // the whole wave at once: a lookup sees the window up to the start of its wave
                                el = tl[hl[j]]; es = ts[hs[j]];
                                const unsigned short val = (unsigned short)(V + 1);
                                tl[hl[j]] = val; ts[hs[j]] = val;
                            }
                        }
                        if (ph < 3) lds_barrier();
                    }
                } else if (can_match[j]) {
                    word_l[j] = S.tbl_long[hl[j] >> 1];
                    word_s[j] = S.tbl_short[hs[j] >> 1];
                    el = (word_l[j] >> ((hl[j] & 1) * 16)) & 0xFFFF; es = (word_s[j] >> ((hs[j] & 1) * 16)) & 0xFFFF;
                }
                if (can_match[j]) {
                    if (search) {
                        const u32 dl = (V + 1 - el) & 0xFFFF, ds = (V + 1 - es) & 0xFFFF;  // distance to the nearest position the entry can mean
                        have_r[j] = rep && V + reach_sec >= rep;
                        have_f[j] = FAR && far_off[j] != 0 && far_off[j] != rep && p + reach_sec >= far_off[j];
                        have_l[j] = el != 0 && dl != 0 && dl <= V;
                        have_s[j] = es != 0 && ds != 0 && ds <= V && short_in_reach<FAR>(ds);
                        if (have_r[j]) q_r[j] = V - rep;
                        if (have_l[j]) q_l[j] = V - dl;
                        if (have_s[j]) q_s[j] = V - ds;
                        // (a candidate whose 8 bytes would straddle the end of the dictionary is dropped)
                        if (q_r[j] < D && q_r[j] + 8 > D) { have_r[j] = false; q_r[j] = V; }
                        if (q_l[j] < D && q_l[j] + 8 > D) { have_l[j] = false; q_l[j] = V; }
                        if (q_s[j] < D && q_s[j] + 8 > D) { have_s[j] = false; q_s[j] = V; }
                    }
                }
            }
            TR(1);
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                w_r[j] = w_l[j] = w_s[j] = 0;
                if (can_match[j] && search) {
                    // only lanes that have a candidate fetch it, and a position named twice is fetched once: the
                    // gathers are what this kernel waits for (one line request per active lane)
                    const bool same_lr = have_l[j] && have_r[j] && q_l[j] == q_r[j];
                    const bool same_sr = have_s[j] && have_r[j] && q_s[j] == q_r[j];
                    const bool same_sl = have_s[j] && have_l[j] && q_s[j] == q_l[j];
                    if (have_r[j]) w_r[j] = ld64(at(q_r[j]));
                    if (have_l[j] && !same_lr) w_l[j] = ld64(at(q_l[j]));
                    if (have_s[j] && !same_sr && !same_sl) w_s[j] = ld64(at(q_s[j]));
                    if (FAR && have_f[j]) w_f[j] = ld64(at(ip + t + ENC_THREADS * j - far_off[j]));
                    if (same_lr) w_l[j] = w_r[j];
                    if (same_sr) w_s[j] = w_r[j];
                    else if (same_sl) w_s[j] = w_l[j];
                }
            }
            {   // the next window's own bytes travel together with this window's candidates (unless S.wbuf will hold them)
                const u32 pn = ip + WIN + t;
                if (!wb_next) {
                    pref_ip = ip + WIN;
                    v_pref = 0;
                    if (pn + 8 <= n) v_pref = ld64(src + pn);
                    else if (pn < n) v_pref = src[pn];
                }
                if (FAR) { hint_ip = ip + WIN; hint_pref = pn < n ? hints[pn >> ZL_FAR_HINT_LOG] : 0u; }
            }
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                const u32 p = ip + t + ENC_THREADS * j;
                if (can_match[j] && search) {
                    const u32 l_r = common_bytes(v[j], w_r[j]), l_l = common_bytes(v[j], w_l[j]), l_s = common_bytes(v[j], w_s[j]);
                    if (have_r[j] && l_r >= MIN_MATCH) { best_len[j] = l_r; best_off[j] = rep; q_best[j] = q_r[j]; }
                    if (have_l[j] && l_l >= tmin && l_l > best_len[j]) { best_len[j] = l_l; best_off[j] = D + p - q_l[j]; q_best[j] = q_l[j]; }
                    if (have_s[j] && best_len[j] < 8 && l_s >= tmin && l_s > best_len[j]) { best_len[j] = l_s; best_off[j] = D + p - q_s[j]; q_best[j] = q_s[j]; }
                    // the far offset: worth its longer code only when all eight bytes agree and nothing nearer does as well
                    if (FAR && have_f[j] && best_len[j] < 8 && common_bytes(v[j], w_f[j]) == 8) { best_len[j] = 8; best_off[j] = far_off[j]; q_best[j] = p - far_off[j]; }
                    TR(2);
                    // extend while whole 8-byte words keep matching
                    if (wb_ok) {  // sixteen bytes per round, the own words from S.wbuf (never a dictionary frame)
                        static_assert((lane_cap - 8) % 16 == 0, "8 + 16 k");
                        while (best_len[j] >= 8 && (best_len[j] & 7) == 0 && best_len[j] < lane_cap) {
                            const u32 l = wb_common16(t + best_len[j], at(q_best[j] + best_len[j]));
                            best_len[j] += l;
                            if (l < 16) break;
                        }
                    } else
                    while (best_len[j] >= 8 && (best_len[j] & 7) == 0 && best_len[j] < lane_cap && p + best_len[j] + 8 <= n) {
                        const u32 qv = q_best[j] + best_len[j];
                        if (qv < D && qv + 8 > D) break;  // the source would run off the end of the dictionary
                        const u32 l = common_bytes(ld64(src + p + best_len[j]), ld64(at(qv)));
                        best_len[j] += l;
                        if (l < 8) break;
                    }
                }
            }
            bool has_match[PPT];
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                has_match[j] = best_len[j] >= MIN_MATCH && worth_it<FAR>(best_len[j], best_off[j], rep);
                S.cand[t + ENC_THREADS * j] = has_match[j] ? (best_len[j] | (best_off[j] << 8)) : 0u;
            }
            // (for the backward extension behind the parse: the eight bytes in front of the match's source are asked for now,
            // by every lane that holds a match, so that they travel during the parse; the eight bytes in front of the
            // position itself are what the lane eight positions down holds)
            u64 back_theirs = 0, back_mine = 0;
            bool back_ok = false;
            if (back_on) {
                const u32 p = ip + t, q = q_best[0];
                // (one frame: the source must stay inside the chunk's section -- sections are self-contained, zl_encode.h)
                const bool src_ok = FAR ? ((i32)q + (i32)reach_sec >= 8) : (DICT ? (q >= D + 8 || (q <= D && q >= 8)) : q >= 8);
                // (every lane that holds a match: whether the byte in front of it is a free literal is known only behind the
                // lazy rule and the parse; asking only lanes whose lower neighbour holds no match lost most of the gain --
                // a neighbour's shorter match that the lazy rule drops is exactly the case in which the extension pays)
                back_ok = has_match[0] && (t & 63) >= 8 && p >= 8 && src_ok;
                if (back_ok) back_theirs = ld64(at(q - 8));
                const u32 lo = __shfl_up((u32)v[0], 8, 64), hi = __shfl_up((u32)(v[0] >> 32), 8, 64);
                back_mine = (u64)lo | ((u64)hi << 32);
            }
            TR(3);
            const u64 s1 = prof ? __builtin_amdgcn_s_memtime() : 0;
            lds_barrier();
            if (wb_ok) wb_advance();
            TR(4);
            // ---- one-step lazy evaluation (cf. the ip+1 long-match check of zstd.c:30025-30040): if the
            // next position holds a clearly better match, emit this byte as a literal instead
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                const u32 w = t + ENC_THREADS * j;
                if (has_match[j] && w + 1 < WIN) {
                    const u32 nx = S.cand[w + 1];
                    if (nx) {
                        const int g1 = 4 * (int)best_len[j] - (int)(31 - __clz(best_off[j] + 1)) + 4;
                        const int g2 = 4 * (int)(nx & 0xFF) - (int)(31 - __clz((nx >> 8) + 1));
                        if (g2 > g1) has_match[j] = false;
                    }
                }
                // levels >= 6: ... and two positions on (zstd.c:31546 ZSTD_compressBlock_lazy_generic at depth 2 asks for a gain of 7 there)
                if (a.lazy2 && has_match[j] && w + 2 < WIN) {
                    const u32 nx = S.cand[w + 2];
                    if (nx) {
                        const int g1 = 4 * (int)best_len[j] - (int)(31 - __clz(best_off[j] + 1)) + 7;
                        const int g2 = 4 * (int)(nx & 0xFF) - (int)(31 - __clz((nx >> 8) + 1));
                        if (g2 > g1) has_match[j] = false;
                    }
                }
            }
            // ---- greedy parse over the window.  Every wave owns 64 consecutive positions: it closes its own
            // (position -> next position) links by pointer jumping through lane shuffles, which tells for
            // every possible entry point where the path leaves the segment; the four segments are then
            // chained by three dependent LDS reads, and each wave marks its on-path positions by replaying
            // its jumps.  One workgroup barrier instead of one per doubling round.
            static_assert(PPT == 1 && WIN == 256, "the wave-local parse assumes one position per thread and four waves");
            const u32 wbase = t & ~63u, wave = t >> 6;
            u32 nxt[PPT];
            nxt[0] = t + (has_match[0] ? best_len[0] : 1u);
            u32 jr[6];
            u32 jp = nxt[0];
    #pragma unroll
            for (int r = 0; r < 6; r++) {
                jr[r] = jp;
                const u32 far = __shfl(jp, (int)((jp - wbase) & 63), 64);
                jp = (jp < wbase + 64) ? far : jp;
            }
            S.jump[0][t] = (u16)jp;  // where a path entering at t leaves this wave's segment
            TR(5);
            lds_barrier();
            TR(6);
            u32 enter = carry, my_entry = 0xFFFFFFFFu;
    #pragma unroll
            for (u32 k = 0; k < 4; k++) {
                if (enter < 64 * (k + 1)) {
                    if (k == wave) my_entry = enter;
                    enter = S.jump[0][enter];
                }
            }
            const u32 next_carry = enter - WIN;  // uniform: where the path enters the next window
            TR(7);
            const bool on_path0 = zl_path_marks(jr, t, wbase, my_entry);
            TR(8);
            const u64 s2 = prof ? __builtin_amdgcn_s_memtime() : 0;
            bool is_seq[PPT], is_lit[PPT], on_path[PPT];
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                on_path[j] = on_path0;
                is_seq[j] = on_path[j] && has_match[j];
                is_lit[j] = on_path[j] && !has_match[j] && in_range[j];
            }
            // ---- a match usually began a few bytes in front of the position whose hash found it (its first bytes hashed to
            // entries that had been overwritten, or lay inside the window's blind spot): while the byte in front of the match
            // equals the byte in front of its source and is still a free literal of this wave, the sequence starts there
            // (zstd.c:29979 "catch up").  CPU model of the parse: text +3.3 %, python sources +1.7 % of ratio; on the device
            // 64 MiB of text in one frame 4.447 -> 4.604 (the reference's level 3: 4.512), the sources 4.316 -> 4.374 (4.428).
            u32 back[PPT];
            back[0] = 0;
            if (back_on) {
                const u64 ms0 = __ballot(is_seq[0]), ml0 = __ballot(is_lit[0]);
                const u32 lane = t & 63;
                u32 k = 0;
                if (is_seq[0] && back_ok) {
                    // free literals right in front of this lane, inside the wave
                    const u64 below_inv = ~ml0 & ((1ull << lane) - 1);                         // lanes below that are NOT literals
                    const u32 run = below_inv ? lane - (64u - (u32)__builtin_clzll(below_inv)) : lane;  // distance to the nearest of them, less one
                    const u64 x = back_mine ^ back_theirs;
                    k = x ? (u32)(__builtin_clzll(x) >> 3) : 8u;                                // equal bytes counted from the end
                    k = k < run ? k : run;
                    k = k < ZL_BACK_MAX ? k : ZL_BACK_MAX;
                }
                // a literal lane asks the next sequence lane above it how far back that one reaches
                const u64 above = lane == 63 ? 0ull : (ms0 >> (lane + 1));
                const u32 nxt_seq = above ? lane + 1 + (u32)__builtin_ctzll(above) : lane;
                const u32 k_nxt = __shfl(k, (int)nxt_seq, 64);
                if (is_lit[0] && above && k_nxt >= nxt_seq - lane) is_lit[0] = false;
                back[0] = k;
            }
            // compaction ranks: the flags are single bits, so a wave ranks its lanes with two ballots and
            // the four waves exchange their totals through LDS
            u32 ex_s[PPT], ex_l[PPT], tot_s, tot_l;
            {
                const u64 ms = __ballot(is_seq[0]), ml = __ballot(is_lit[0]);
                const u64 below = (1ull << (t & 63)) - 1;
                const u32 ws = (u32)__popcll(ms), wl = (u32)__popcll(ml);
                // (one frame: how many of the wave's sequences have an offset below 64 rides along in the top bits)
                const u32 wn = (FAR && INTRA && !HI) ? (u32)__popcll(__ballot(is_seq[0] && best_off[0] < 64)) : 0u;
                if ((t & 63) == 0) S.scan_a[wave] = ws | (wl << 16) | (wn << 25);
                lds_barrier();
                u32 pre = 0, tot = 0, nshort = 0;
    #pragma unroll
                for (u32 w = 0; w < 4; w++) {
                    const u32 xx = S.scan_a[w], x = xx & 0x1FFFFFFu;
                    if (w < wave) pre += x;
                    tot += x;
                    nshort += xx >> 25;
                }
                ex_s[0] = (pre & 0xFFFF) + (u32)__popcll(ms & below);
                ex_l[0] = (pre >> 16) + (u32)__popcll(ml & below);
                tot_s = tot & 0xFFFF;
                tot_l = tot >> 16;
                // the next window looks up and inserts eight lanes at a time while a quarter of this one's sequences repeat at a
                // distance below 64 (records of 8 or 16 bytes, the next line of a listing); a pure function of the data
                if (FAR && INTRA && !HI && tot_s >= 4) sub8 = nshort * ZL_SUB8_DIV >= tot_s;  // (levels >= 5 keep the sub-phases throughout: they asked for ratio)
            }
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                const u32 p = ip + t + ENC_THREADS * j;
                if (is_lit[j]) lits[nlit + ex_l[j]] = (u8)v[j];
                if (is_seq[j]) {
                    zl_seq sq;
                    sq.lit_cum = nlit + ex_l[j];
                    sq.match_len = best_len[j] + back[j];
                    sq.offset = best_off[j];
                    sq.pos = p - back[j];
                    seqs[nseq + ex_s[j]] = sq;
                    if (ex_s[j] + 1 == tot_s) S.rep = best_off[j];  // last sequence of the window
                }
            }
            carry = next_carry;
            TR(10);
            const u64 s3 = prof ? __builtin_amdgcn_s_memtime() : 0;
            // ---- insert every position of the window (most recent position wins)
    #pragma unroll
            for (int j = 0; j < PPT; j++) {
                if (can_match[j]) {
                    const u32 p = ip + t + ENC_THREADS * j;
                    if (ZL_INSERT_CAS) insert16_pair(S.tbl_long, hl[j], word_l[j], S.tbl_short, hs[j], word_s[j], D + p, D + ip, WIN);
                }
            }
            if (!ZL_INSERT_CAS && !(INTRA && intra_on)) {  // (the look-up-and-insert phases above have done it)
                for (u32 ph = 0; ph < 4; ph++) {
                    if (ph == (t >> 6) && can_match[0]) {
                        const unsigned short val = (unsigned short)(D + ip + t + 1);
                        ((unsigned short*)S.tbl_long)[hl[0]] = val;
                        ((unsigned short*)S.tbl_short)[hs[0]] = val;
                    }
                    if (ph < 3) lds_barrier();
                }
            }
            ZL_LITS_CATCH_UP(nlit);
            nseq += tot_s;
            nlit += tot_l;
            win_seqs = tot_s;
            win_lits = tot_l;
            TR(11);
            lds_barrier();
            TR(12);
#ifdef ZL_TRACE
            if (tr_on && tr_it < ZL_TRACE_ITERS) a.stats[16 + (tr_it * 4 + (threadIdx.x >> 6)) * ZL_TRACE_POINTS + 9] = a.stats[16 + (tr_it * 4 + (threadIdx.x >> 6)) * ZL_TRACE_POINTS + 10];
            tr_it++;
#endif
            if (prof) {
                const u64 s4 = __builtin_amdgcn_s_memtime();
                c_find += s1 - s0; c_parse += s2 - s1; c_emit += s3 - s2; c_ins += s4 - s3; c_it++;
            }

        } else {
            win = WIN2;
            // the stride-2 parse runs over even positions: an odd entry point is one literal first
            const bool covered = carry >= WIN2;
            if (!covered && (carry & 1)) {
                if (ip + carry < n) {
                    if (t == 0) lits[nlit] = src[ip + carry];
                    nlit += 1;
                }
                carry += 1;
            }
            const u32 carry_n = carry >> 1;
            const u32 wbase = t & ~63u, wave = t >> 6;
            const u64 s0 = prof ? __builtin_amdgcn_s_memtime() : 0;
            const u32 rep = S.rep;
            const u32 p = ip + 2 * t;
            const bool in0 = p < n, in1 = p + 1 < n;
            const bool can0 = p + 8 <= n, can1 = p + 9 <= n;  // the last 7 bytes of a chunk are always literals
            // 16 bytes from p - 1: the byte in front of p (for extending a match backwards), the 8 bytes at p, the 8 at p + 1
            u64 lo = 0, hi = 0;
            if (pref_ip2 == ip) { lo = pref0; hi = pref1; }  // fetched while the previous window was being searched
            else if (p >= 1 && p + 15 <= n) { lo = ld64(src + p - 1); hi = ld64(src + p + 7); }
            else { for (u32 i = 0; i < 8; i++) { if (p + i >= 1 && p + i - 1 < n) lo |= (u64)src[p + i - 1] << (8 * i); if (p + 7 + i < n) hi |= (u64)src[p + 7 + i] << (8 * i); } }
            const u32 byte_before = (u32)lo & 0xFF;
            const u64 v0 = (lo >> 8) | (hi << 56), v1 = (lo >> 16) | (hi << 48);
            const bool search = carry_n < ENC_THREADS;  // otherwise the whole window lies inside an earlier match
            u32 hl0 = 0, hs0 = 0, hl1 = 0, hs1 = 0, wl0 = 0, ws0 = 0, wl1 = 0, ws1 = 0;
            u32 best_len = 0, best_off = 0, q_best = p;
            bool have_r = false, have_l = false, have_s = false, have_f = false;
            u32 q_r = p, q_l = p, q_s = p;
            u32 far_off2 = 0;
            if (FAR) {
                const u32 hw = (pref_ip2 == ip) ? hint_pref : (in0 ? hints[p >> ZL_FAR_HINT_LOG] : 0u);
                far_off2 = (smin > ZL_SPARSE_MIN && (hw >> 26) < ZL_CHEAP_FAR_MIN) ? 0u : (hw & 0x3FFFFFFu);  // (cheap bytes: a far offset costs 30 bits)
            }
            u32 el = 0, es = 0;
            if (can0) { hl0 = hash_long(v0, lshift); hs0 = hash_short_rt(v0, sbsh, sshift); }
            if (can1) { hl1 = hash_long(v1, lshift); hs1 = hash_short_rt(v1, sbsh, sshift); }
            if (INTRA && intra_on) {  // look up and insert, wave after wave (see the dense body); both positions of every lane go in
                unsigned short* const tl = (unsigned short*)S.tbl_long;
                unsigned short* const ts = (unsigned short*)S.tbl_short;
                for (u32 ph = 0; ph < 4; ph++) {
                    if (ph == wave) {
                        if (can0) { el = tl[hl0]; es = ts[hs0]; }
                        if (can0) { tl[hl0] = (unsigned short)(p + 1); ts[hs0] = (unsigned short)(p + 1); }
                        if (can1) { tl[hl1] = (unsigned short)(p + 2); ts[hs1] = (unsigned short)(p + 2); }
                    }
                    if (ph < 3) lds_barrier();
                }
            } else {
                if (can0) { wl0 = S.tbl_long[hl0 >> 1]; ws0 = S.tbl_short[hs0 >> 1]; el = (wl0 >> ((hl0 & 1) * 16)) & 0xFFFF; es = (ws0 >> ((hs0 & 1) * 16)) & 0xFFFF; }
                if (can1) { wl1 = S.tbl_long[hl1 >> 1]; ws1 = S.tbl_short[hs1 >> 1]; }  // the odd position is only inserted: its table words are read here, next to the others
            }
            if (can0) {
                if (search) {
                    const u32 dl = (p + 1 - el) & 0xFFFF, ds = (p + 1 - es) & 0xFFFF;  // distance to the nearest position the entry can mean
                    have_r = rep && p + reach_sec >= rep;
                    have_f = FAR && far_off2 != 0 && far_off2 != rep && p + reach_sec >= far_off2;
                    have_l = el != 0 && dl != 0 && dl <= p;
                    have_s = es != 0 && ds != 0 && ds <= p && short_in_reach<FAR>(ds);
                    if (have_r) q_r = p - rep;
                    if (have_l) q_l = p - dl;
                    if (have_s) q_s = p - ds;
                }
            }
            u64 w_r = 0, w_l = 0, w_s = 0, w_f = 0;
            if (can0 && search) {  // (as in the stride-1 body: candidates are fetched by the lanes that have one, each position once)
                const bool same_lr = have_l && have_r && q_l == q_r, same_sr = have_s && have_r && q_s == q_r, same_sl = have_s && have_l && q_s == q_l;
                if (have_r) w_r = ld64(at(q_r));
                if (have_l && !same_lr) w_l = ld64(src + q_l);
                if (have_s && !same_sr && !same_sl) w_s = ld64(src + q_s);
                if (FAR && have_f) w_f = ld64(at(p - far_off2));
                if (same_lr) w_l = w_r;
                if (same_sr) w_s = w_r;
                else if (same_sl) w_s = w_l;
            }
            {   // the next window's own bytes travel together with this window's candidates
                const u32 pn = ip + WIN2 + 2 * t;
                pref_ip2 = ip + WIN2;
                pref0 = 0; pref1 = 0;
                if (pn + 15 <= n) { pref0 = ld64(src + pn - 1); pref1 = ld64(src + pn + 7); if (FAR) hint_pref = hints[pn >> ZL_FAR_HINT_LOG]; }
                else pref_ip2 = 0xFFFFFFFFu;  // the ragged end is read byte by byte when it is reached
            }
            if (can0 && search) {
                const u32 l_r = common_bytes(v0, w_r), l_l = common_bytes(v0, w_l), l_s = common_bytes(v0, w_s);
                if (have_r && l_r >= MIN_MATCH) { best_len = l_r; best_off = rep; q_best = q_r; }
                // (a 4-byte hit is kept for now: it is worth a new offset only if the byte in front matches too, see below)
                if (have_l && l_l >= MIN_MATCH && l_l > best_len) { best_len = l_l; best_off = p - q_l; q_best = q_l; }
                if (have_s && best_len < 8 && l_s >= MIN_MATCH && l_s > best_len) { best_len = l_s; best_off = p - q_s; q_best = q_s; }
                if (FAR && have_f && best_len < 8 && common_bytes(v0, w_f) == 8) { best_len = 8; best_off = far_off2; q_best = p - far_off2; }
                // extend while whole 8-byte words keep matching
                while (best_len >= 8 && (best_len & 7) == 0 && best_len < lane_cap && p + best_len + 8 <= n) {
                    const u32 l = common_bytes(ld64(src + p + best_len), ld64(at(q_best + best_len)));
                    best_len += l;
                    if (l < 8) break;
                }
            }
            bool has_match = best_len >= MIN_MATCH && worth_it<FAR>(best_len, best_off, rep);
            // A repeat that began on the odd position in front of p is found here one byte late.  The byte in
            // front of the candidate is fetched now (it is needed when the sequence is written): if it
            // equals the byte in front of p and that byte is still a free literal, the match starts there.
            u32 cand_before = 0x100;
            if (has_match && p >= 1 && (FAR ? (i32)q_best + (i32)reach_sec >= 1 : q_best >= 1)) cand_before = *at(q_best - 1);  // (inside the chunk's SECTION: the byte in front of a section is not this launch's to read -- a pledged stream's second job holds its own section only, and sections are self-contained)
            const bool back_ok = cand_before == byte_before;
        if (has_match && best_off != rep && best_len + (back_ok ? 1u : 0u) < smin) has_match = false;
            S.cand[t] = has_match ? (best_len | (best_off << 8)) : 0u;
            // does any lane of the workgroup hold a match?  (one word per wave next to the candidates, read behind the same barrier)
            {
                const u64 many = __ballot(has_match);
                if ((t & 63) == 0) S.scan_b[t >> 6] = many ? 1u : 0u;
            }
            const u64 s1 = prof ? __builtin_amdgcn_s_memtime() : 0;
            u64 s2 = s1;
            lds_barrier();
            const bool any_match = (S.scan_b[0] | S.scan_b[1] | S.scan_b[2] | S.scan_b[3]) != 0;  // uniform
            u32 tot_s = 0, tot_l = 0;
            if (!any_match) {
                // ---- a window without a single match -- the rule on data that does not repeat -- needs no parse: every node from
                // the entry point on is on the path and gives its two bytes (what is left of them at the end of the chunk)
                if (!covered && t >= carry_n && nseq != 0) {  // (a chunk without a sequence so far keeps its literals where they are: see lits_pending below)
                    const u32 ex = 2 * (t - carry_n);
                    if (in0) lits[nlit + ex] = (u8)v0;
                    if (in1) lits[nlit + ex + 1] = (u8)(v0 >> 8);
                }
                if (!covered && ip + 2 * carry_n < n) {
                    const u32 left = n - (ip + 2 * carry_n), span = 2 * (ENC_THREADS - carry_n);
                    tot_l = left < span ? left : span;
                }
                carry = covered ? carry - WIN2 : 0u;
            } else {
                // ---- lazy evaluation against the next probed position (two bytes on): giving up this match costs two literals
                if (has_match && t + 1 < ENC_THREADS) {
                    const u32 nxc = S.cand[t + 1];
                    if (nxc) {
        #ifndef ZL_LAZY_BIAS
        #define ZL_LAZY_BIAS 8
        #endif
                        const int g1 = 4 * (int)best_len - (int)(31 - __clz(best_off + 1)) + ZL_LAZY_BIAS;
                        const int g2 = 4 * (int)(nxc & 0xFF) - (int)(31 - __clz((nxc >> 8) + 1));
                        if (g2 > g1) has_match = false;
                    }
                }
                // the byte behind a match that ends on an odd position is a literal of this lane
                const u32 mend = 2 * t + best_len;  // window-relative end of the match
                const bool odd_end = has_match && (mend & 1) && p + best_len < n;
                u32 tail_byte = 0;
                // (the byte behind a match of up to 14 bytes is one of the lane's own sixteen; a load here is a memory round trip in the parse)
                if (odd_end) {
                    if (best_len <= 14) { const u32 idx = best_len + 1; tail_byte = (u32)((idx < 8 ? lo : hi) >> (8 * (idx & 7))) & 0xFF; }
                    else tail_byte = src[p + best_len];
                }
                // ---- greedy parse over the 256 nodes (even positions): wave-local closure, chain, marks
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
                u32 enter = carry_n, my_entry = 0xFFFFFFFFu;
        #pragma unroll
                for (u32 k = 0; k < 4; k++) {
                    if (enter < 64 * (k + 1)) {
                        if (k == wave) my_entry = enter;
                        enter = S.jump[0][enter];
                    }
                }
                const u32 next_carry = enter - ENC_THREADS;
                const bool on_path = zl_path_marks(jr, t, wbase, my_entry);
                s2 = prof ? __builtin_amdgcn_s_memtime() : 0;
                const bool is_seq = on_path && has_match;
                // backward extension by one byte.  The byte in front of an on-path node is a literal owned by the
                // node before it on the path: the odd byte of a lane without a match, or the byte behind a match
                // that ends on an odd position.  The wanting lane raises a flag in its slot, the owner answers
                // in the second array; both then count one literal less / one match byte more.
                const bool givable = on_path && nxt < ENC_THREADS && (has_match ? odd_end : in1);
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
                u32 ex_s, ex_l;
                {
                    const u64 ms = __ballot(is_seq), m1 = __ballot((nl_mine & 1) != 0), m2 = __ballot((nl_mine & 2) != 0);
                    const u64 below = (1ull << (t & 63)) - 1;
                    const u32 ws = (u32)__popcll(ms), wl = (u32)__popcll(m1) + 2 * (u32)__popcll(m2);
                    if ((t & 63) == 0) S.scan_a[wave] = ws | (wl << 16);
                    lds_barrier();
                    u32 pre = 0, tot = 0;
        #pragma unroll
                    for (u32 w = 0; w < 4; w++) {
                        const u32 x = S.scan_a[w];
                        if (w < wave) pre += x;
                        tot += x;
                    }
                    ex_s = (pre & 0xFFFF) + (u32)__popcll(ms & below);
                    ex_l = (pre >> 16) + (u32)__popcll(m1 & below) + 2 * (u32)__popcll(m2 & below);
                    tot_s = tot & 0xFFFF;
                    tot_l = tot >> 16;
                }
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
                carry = covered ? carry - WIN2 : 2 * next_carry;
            }
            const u64 s3 = prof ? __builtin_amdgcn_s_memtime() : 0;
            // ---- insert both positions of every lane (most recent position wins)
            if (ZL_INSERT_CAS) {
                if (can0) insert16_pair(S.tbl_long, hl0, wl0, S.tbl_short, hs0, ws0, p, ip, WIN2);
                if (can1) insert16_pair(S.tbl_long, hl1, wl1, S.tbl_short, hs1, ws1, p + 1, ip, WIN2);
            } else if (!(INTRA && intra_on)) {
                for (u32 ph = 0; ph < 4; ph++) {
                    if (ph == wave) {
                        if (can0) { ((unsigned short*)S.tbl_long)[hl0] = (unsigned short)(p + 1); ((unsigned short*)S.tbl_short)[hs0] = (unsigned short)(p + 1); }
                        if (can1) { ((unsigned short*)S.tbl_long)[hl1] = (unsigned short)(p + 2); ((unsigned short*)S.tbl_short)[hs1] = (unsigned short)(p + 2); }
                    }
                    if (ph < 3) lds_barrier();
                }
            }
            ZL_LITS_CATCH_UP(nlit);
            nseq += tot_s;
            nlit += tot_l;
            win_seqs = tot_s;
            win_lits = tot_l;
            lds_barrier();
            if (prof) {
                const u64 s4 = __builtin_amdgcn_s_memtime();
                c_find += s1 - s0; c_parse += s2 - s1; c_emit += s3 - s2; c_ins += s4 - s3; c_it++;
            }

        }
        // ---- search acceleration on data without matches (cf. the growing step of zstd.c:30018-30023):
        // after 4 / 8 / 12 ... consecutive windows that produced no sequence, the next 1 / 2 / 3 ... (at most 7) windows are
        // emitted as literals without searching or inserting.  A pure function of the data, so the
        // output stays deterministic.  Any sequence resets it.
        miss = (win_seqs == 0 && carry == 0) ? miss + 1 : 0;
        // (a dull chunk -- see above -- ramps up twice as fast and skips up to 15 windows: the reference's step on such data grows with
        // every 256 bytes without a match, zstd.c:30018, and reaches hundreds; 1 GiB of random bytes 0.80 -> 0.55 ms, of i.i.d. bytes over ten letters 1.80 -> 1.30 ms, same ratios)
        u32 skip = (miss * (1u + a.accel)) >> (dull ? 1 : 2);  // (negative levels: -accel)
#ifndef ZL_SKIP_MAX
#define ZL_SKIP_MAX 7
#endif
        if (skip > (dull ? 2u * ZL_SKIP_MAX + 1u : (u32)ZL_SKIP_MAX)) skip = dull ? 2u * ZL_SKIP_MAX + 1u : (u32)ZL_SKIP_MAX;
        ip += win;
        if (skip && ip < n) {
            const u32 to = (ip + skip * win < n) ? ip + skip * win : n;
            if (nseq != 0) {   // 16 bytes per thread and trip (global memory takes unaligned vectors)
                const u32 cnt = to - ip, body = cnt & ~15u;
                for (u32 q = t * 16; q < body; q += ENC_THREADS * 16) { uint4 x; __builtin_memcpy(&x, src + ip + q, 16); __builtin_memcpy(lits + nlit + q, &x, 16); }
                if (t < cnt - body) lits[nlit + body + t] = src[ip + body + t];
            }
            nlit += to - ip;
            ip = to;
        }
#if ZL_DENSE_LIT_PCT_FAR
        // windowed layout: the dense body where MATCHES COVER the data (literals at most this share of the window), not where
        // sequences are many: i.i.d. bytes over a small alphabet yield twenty 5-byte matches per 512 bytes that are worth
        // little (iid10 in one frame: 5.7 ms per 256 MiB in the dense body for a WORSE ratio than the sparse one's)
        if (FAR) dense = !a.sparse_only && win_lits * 100u <= win * (u32)ZL_DENSE_LIT_PCT_FAR;
        else
#endif
        dense = D != 0 || (!a.sparse_only && win_seqs * (WIN2 / win) >= (FAR ? ZL_DENSE_SEQS_FAR : ZL_DENSE_SEQS));
        if ((cheap || noisy) && probed < 4096) {
            probed += win; probed_lits += win_lits;
            if (probed >= 4096 && (probed - probed_lits) * 10 < probed) { if (cheap) smin = 8; dull = true; }
        }
    }
#undef tmin
#undef ZL_LITS_CATCH_UP
    if (prof) {
        atomicAdd((unsigned long long*)&a.stats[0], (unsigned long long)c_find);
        atomicAdd((unsigned long long*)&a.stats[1], (unsigned long long)c_parse);
        atomicAdd((unsigned long long*)&a.stats[2], (unsigned long long)c_emit);
        atomicAdd((unsigned long long*)&a.stats[3], (unsigned long long)c_ins);
        atomicAdd((unsigned long long*)&a.stats[4], (unsigned long long)c_it);
        atomicAdd((unsigned long long*)&a.stats[5], (unsigned long long)nseq);
    }
    // ---- merge continuation pieces.  A lane verifies at most LANE_MATCH_CAP bytes, so a longer match arrives as
    // several sequences: one starts where the last one ends, with the same offset and no literal between them.
    // Coded as they are, every piece after the first would spell its offset out (a repeat offset needs a literal in
    // front of it, zstd.c:19680); joined, they are the one sequence the reference would have found.  One pass over
    // the chunk's sequences, 256 at a time, compacting in place: heads move down, pieces add their length to
    // their head.  (The match tables are dead by now: their LDS holds the tile's ends and offsets.)
    const u32 out_n = zl_merge_pieces(seqs, nseq, S.tbl_long, S.tbl_long + 256, S.scan_a);  // (the match tables are dead by now: their LDS holds the tile's ends and offsets)
    if (t == 0) {
        a.meta[c].nseq = out_n;
        a.meta[c].nlit = nlit;
    }
}

// The match tables as they look after the last byte of a dictionary's content: one workgroup walks the
// content window by window with the same ordered insert as the encoder (run once per loaded dictionary).
__global__ __launch_bounds__(ENC_THREADS) void zl_dict_hash_kernel(const u8* content, u32 n, u32* out) {
    __shared__ MatchSharedT<LONG_LOG, SHORT_LOG> S;
    const u32 t = threadIdx.x;
    for (u32 i = t; i < (1u << LONG_LOG) / 2; i += ENC_THREADS) S.tbl_long[i] = 0;
    for (u32 i = t; i < (1u << SHORT_LOG) / 2; i += ENC_THREADS) S.tbl_short[i] = 0;
    lds_barrier();
    for (u32 ip = 0; ip + 8 <= n; ip += ENC_THREADS) {
        const u32 v_pos = ip + t;
        const bool can = v_pos + 8 <= n;
        u32 hl = 0, hs = 0;
        if (can) { const u64 v = ld64(content + v_pos); hl = hash8(v); hs = hash5(v); }
        for (u32 ph = 0; ph < 4; ph++) {
            if (ph == (t >> 6) && can) {
                const unsigned short val = (unsigned short)(v_pos + 1);
                ((unsigned short*)S.tbl_long)[hl] = val;
                ((unsigned short*)S.tbl_short)[hs] = val;
            }
            lds_barrier();
        }
    }
    for (u32 i = t; i < (1u << LONG_LOG) / 2; i += ENC_THREADS) out[i] = S.tbl_long[i];
    for (u32 i = t; i < (1u << SHORT_LOG) / 2; i += ENC_THREADS) out[(1u << LONG_LOG) / 2 + i] = S.tbl_short[i];
}

// ---------------------------------------------------------------------------------------
// Far matches (windowed layout).  The chunk-local tables see 64 KiB at most and nothing in front of the
// chunk; the reference's level 3 sees 2 MiB (zstd.c:29529).  What lies further back is found through a
// hash table in HBM over SAMPLED positions: a position takes part when the top bits of the hash of its
// 8 bytes are zero -- a property of the content, so the two occurrences of a repeat are sampled alike
// (the idea of the reference's long-distance matcher, zstd.c:33790 ZSTD_ldm_*, restated for a machine
// that cannot walk the input in order).  Two passes over the input, both one workgroup per chunk:
//   zl_far_build_kernel  the chunk's COLUMN of the table -- column[bucket] = max (position in chunk << 15 | tag) --
//                        is built in LDS (plain LDS atomics) and written out whole, coalesced: no cell of the table
//                        belongs to two chunks, so nothing needs a device-wide atomic; "max" makes the content
//                        independent of scheduling (determinism).  The sampled positions are also LISTED.
//   zl_far_probe_kernel  streams the columns of the 15 chunks in front through LDS, nearest first; every listed
//                        position (held in registers) looks its bucket up in each, checks a tag hit against the
//                        bytes, and the longest hit of every 16-byte neighbourhood becomes that neighbourhood's
//                        hint word: the offset the match kernel tries there.
// (The first version kept one 64-byte line of 16 chunk slots per bucket, filled with device atomics and read with
// two random line fetches per sampled position: bound by random transactions -- 384 M of them per GiB, 12 ms.
// Columns make every table access but the verification a coalesced stream: 8 GiB of reads per GiB of input.)
// All chunks but the last are ZL_CHUNK bytes and contiguous, so chunk k steps back are k * ZL_CHUNK bytes.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 far_hash(u64 v) { return v * 0x9E3779B185EBCA87ULL; }
// (sl: how many top bits of the hash must be zero for a position to take part: one in 2^sl does -- the level's knob)
__device__ __forceinline__ bool far_sampled(u64 h, u32 sl) { return (h >> (64 - sl)) == 0; }
__device__ __forceinline__ u32 far_bucket(u64 h, u32 sl) { return (u32)(h >> (64 - sl - ZL_FAR_BUCKET_LOG)) & ((1u << ZL_FAR_BUCKET_LOG) - 1); }
__device__ __forceinline__ u32 far_tag(u64 h, u32 sl) { return (u32)(h >> (64 - sl - ZL_FAR_BUCKET_LOG - 15)) & 0x7FFFu; }
// positions p < far_limit have their 8 bytes inside the input and inside the chunk's section
__device__ __forceinline__ u32 far_limit(const zl_far_args& a, const zl_chunk_desc& cd, u32 cg /* chunk index inside the input */) {
    const u64 sec_end = ((u64)((cg >> ZL_FAR_SECTION_LOG) + 1) << ZL_FAR_SECTION_LOG) * ZL_CHUNK;  // (the input starts at source offset 0)
    const u64 left = (a.total < sec_end ? a.total : sec_end) - cd.src_off;
    return left >= (u64)cd.src_size + 8 ? cd.src_size : (left >= 8 ? (u32)(left - 7) : 0u);
}
#define FAR_COL (1u << ZL_FAR_BUCKET_LOG)
#define FAR_UNROLL 4
static_assert(ZL_FAR_BUCKET_LOG + 15 <= 32 - 3, "a list key holds bucket and tag");

// (A workgroup holds a 64 KiB column in LDS, so only two fit a CU: what hides the memory latency is the number of loads
// each thread keeps in flight, not the number of waves.  Every thread takes 8 consecutive positions from ONE 16-byte
// load, four such loads per trip.)
#define FAR_SPAN 8u                     // positions per thread and load
// (1024 threads per chunk: a streaming kernel whose 64 KiB column allows two workgroups per CU wants every wave slot of them;
// with 256 threads it ran at 1.8 TB/s)
#define FAR_BUILD_THREADS 1024u
__global__ __launch_bounds__(FAR_BUILD_THREADS, 8) void zl_far_build_kernel(zl_far_args a, u32 nchunks) {
    __shared__ u32 col[FAR_COL];
    __shared__ u32 wtot[2][FAR_BUILD_THREADS / 64];
    __shared__ u32 count, sparse_count;
    u32 running = 0;  // ordered pass: sampled positions so far (uniform)
    const u32 c = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const zl_chunk_desc cd = a.chunks[c];
    const u8* src = a.src + cd.src_off;
    const u32 lim = far_limit(a, cd, a.chunk_base + c);
    const u32 sl = a.sample_log;
    u32* list = a.lists + (size_t)c * ZL_FAR_LIST_WORDS;
    u32* const sparse = list + 16 + 2 * (size_t)ZL_FAR_LIST_CAP;  // [0] how many listed positions have a tag whose low byte is zero, then the keys of the first ZL_FAR_SPARSE_CAP to arrive (used only when all of them fit: which ones arrive first is not the content's business)
    for (u32 i = t; i < FAR_COL; i += FAR_BUILD_THREADS) col[i] = 0;
    if (t == 0) { count = 0; sparse_count = 0; }
    {   // the chunk's hint words start out empty (the probe kernel raises them with atomicMax)
        uint4* h4 = (uint4*)(a.hints + (size_t)c * ZL_FAR_HINT_WORDS);
        for (u32 i = t; i < ZL_FAR_HINT_WORDS / 4; i += FAR_BUILD_THREADS) h4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // The list: one slot per sampled position.  First pass: slots handed out per wave from an atomic counter, in whatever order
    // the waves reach it -- which positions share a slot range does not matter as long as ALL of them fit.  When a chunk samples
    // more positions than the list holds (long runs, whose positions all hash alike: all of them are sampled or none), WHICH
    // positions are kept must not depend on scheduling (until round 3 it did: the same input could compress to different
    // bytes; scripts/fuzz_encode.py): such a chunk is listed again, in the order of its positions -- a scan over the waves'
    // totals in wave order, one barrier per step -- and keeps the first ZL_FAR_LIST_CAP.  (The scan for every chunk: 3.2 -> 3.45 ms.)
    for (u32 ordered = 0; ordered < 2; ordered++) {
        if (ordered) {
            __syncthreads();  // (count is final, every position is in the column)
            if (count <= ZL_FAR_LIST_CAP) break;
        }
        for (u32 ip = 0; ip < lim; ip += FAR_BUILD_THREADS * FAR_SPAN * FAR_UNROLL) {
            uint4 w[FAR_UNROLL];
#pragma unroll
            for (int k = 0; k < FAR_UNROLL; k++) {
                const u32 p0 = ip + (k * FAR_BUILD_THREADS + t) * FAR_SPAN;
                w[k] = make_uint4(0, 0, 0, 0);
                if (p0 + 16 <= lim + 7) w[k] = ld128(src + p0);               // 16 bytes from p0 lie inside what far_limit guards (lim + 7 bytes are readable)
                else for (u32 i = 0; i < 16 && p0 + i < lim + 7; i++) ((u32*)&w[k])[i >> 2] |= (u32)src[p0 + i] << (8 * (i & 3));
            }
#pragma unroll
            for (int k = 0; k < FAR_UNROLL; k++) {
                const u32 p0 = ip + (k * FAR_BUILD_THREADS + t) * FAR_SPAN;
                const u64 lo = (u64)w[k].x | ((u64)w[k].y << 32), hi = (u64)w[k].z | ((u64)w[k].w << 32);
                u32 mine = 0, keys[FAR_SPAN];
#pragma unroll
                for (u32 j = 0; j < FAR_SPAN; j++) {
                    const u64 v = j == 0 ? lo : ((lo >> (8 * j)) | (hi << (64 - 8 * j)));
                    const u64 h = far_hash(v);
                    const bool on = p0 + j < lim && far_sampled(h, sl);
                    const u32 bucket = far_bucket(h, sl), tag = far_tag(h, sl);
                    keys[j] = bucket | (tag << ZL_FAR_BUCKET_LOG);
                    if (on) {
                        if (!ordered) {
                            atomicMax(&col[bucket], ((p0 + j) << 15) | tag);
                            if ((tag & 0xFFu) == 0) { const u32 i = atomicAdd(&sparse_count, 1u); if (i < ZL_FAR_SPARSE_CAP) sparse[1 + i] = keys[j]; }
                        }
                        mine |= 1u << j;
                    }
                }
                u32 cnt = (u32)__popc(mine), inc = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const u32 x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
                u32 idx;
                if (!ordered) {
                    const u32 wave_total = __shfl(inc, 63, 64);
                    u32 base = 0;
                    if (lane == 0 && wave_total) base = atomicAdd(&count, wave_total);
                    idx = __shfl(base, 0, 64) + inc - cnt;
                } else {
                    u32* const wt = wtot[k & 1];  // (two sets of totals in turn: a step's totals are read before the step after next writes them again, and the next step's barrier lies in between)
                    if (lane == 63) wt[t >> 6] = inc;
                    __syncthreads();
                    u32 pre = 0, tot = 0;
#pragma unroll
                    for (u32 w2 = 0; w2 < FAR_BUILD_THREADS / 64; w2++) { const u32 x = wt[w2]; pre += w2 < (t >> 6) ? x : 0u; tot += x; }
                    idx = running + pre + inc - cnt;
                    running += tot;
                }
#pragma unroll
                for (u32 j = 0; j < FAR_SPAN; j++) {
                    if ((mine >> j) & 1) {
                        if (idx < ZL_FAR_LIST_CAP) *(uint2*)(list + 16 + 2 * (size_t)idx) = make_uint2(keys[j], p0 + j);
                        idx++;
                    }
                }
            }
        }
    }
    __syncthreads();  // (every position is in the column)
    if (t == 0) { list[0] = count < ZL_FAR_LIST_CAP ? count : ZL_FAR_LIST_CAP; sparse[0] = sparse_count; }
    uint4* out = (uint4*)(a.table + (size_t)c * FAR_COL);
    const uint4* in = (const uint4*)col;
    for (u32 i = t; i < FAR_COL / 4; i += FAR_BUILD_THREADS) out[i] = in[i];
}

// The probe kernel: 1024 threads per PAIR of chunks, sixteen listed positions of each per thread in registers (round 2 held 64
// per thread in a 256-thread workgroup: 256 VGPRs + 13 AGPRs, ONE wave per SIMD, so that nothing overlapped the column loads --
// 4.4 ms per GiB at 1.8 TB/s of column traffic).  One workgroup per CU, two column buffers (128 KiB of LDS).
#define FAR_PROBE_THREADS 1024u
#define FAR_PER_THREAD 16u   // listed positions a thread holds in registers at a time
// Workgroup b of a launch tends to run on XCD b mod 8 (dispatch is round-robin over the XCDs; nothing depends on it).  Runs of
// FAR_XCD_RUN consecutive pairs of chunks go to one XCD: the workgroups resident there at one time then stream the SAME columns (chunk c
// wants those of c-1 .. c-15), and the XCD's L2 serves most of the fifteen-fold re-read that otherwise goes out to the fabric.
// (Runs, not one contiguous share per XCD: the chunks' costs differ with the data, and a share per XCD left most of the device
// idle while the two XCDs that held the text quarter of the bench vector worked: 6.0 -> 8.9 ms.)
#define FAR_XCDS 8u
#ifndef ZL_FAR_GATE
#define ZL_FAR_GATE 1  // a chunk's sampled positions are probed sparsely before the columns are streamed for it (zl_far_probe_kernel)
#endif
#ifndef FAR_XCD_RUN
#define FAR_XCD_RUN 16u
#endif
__device__ __forceinline__ u32 far_xcd_chunk(u32 b) {
#if FAR_XCD_RUN
    const u32 x = b % FAR_XCDS, i = b / FAR_XCDS;
    return ((i / FAR_XCD_RUN) * FAR_XCDS + x) * FAR_XCD_RUN + i % FAR_XCD_RUN;  // (>= nchunks at the ragged end: those workgroups leave)
#else
    return b;
#endif
}
__global__ __launch_bounds__(FAR_PROBE_THREADS, 4) void zl_far_probe_kernel(zl_far_args a, u32 nchunks) {
    // A workgroup serves TWO neighbouring chunks: chunk c wants the columns of c-1 .. c-15, chunk c+1 those of c .. c-14, so a
    // column that is in LDS for one is there for the other and the kernel's column traffic (what it is bound by) halves.  The
    // entries' positions are not held in registers any more (two chunks' keys take their place): a hit keeps offset - position
    // and reads the position again from the list when it is checked.  Thirty-two keys per thread want 128 registers, which is
    // one workgroup per CU; the overlap that two resident workgroups gave each other (one streams while the other looks up)
    // now happens inside the one: two column buffers, the next column's LDS-DMA in flight during the lookups in this one.
    __shared__ __align__(16) u32 colbuf[2][FAR_COL];
    __shared__ u32 still_open[ZL_FAR_GROUP + 2];  // per column step: does any wave hold an entry that is still looking
#if ZL_FAR_GATE
    __shared__ u32 pre_hits[2];
#endif
    const u32 c0 = 2 * far_xcd_chunk(blockIdx.x), t = threadIdx.x;
    if (c0 >= nchunks) return;
    u32 back[2], count[2], lim[2];
    const u32* list[2];
    const u8* src[2];
#pragma unroll
    for (u32 h = 0; h < 2; h++) {
        const u32 c = c0 + h;
        back[h] = 0; count[h] = 0; lim[h] = 0; list[h] = a.lists; src[h] = a.src;
        if (c < nchunks) {
            const zl_chunk_desc cd = a.chunks[c];
            src[h] = a.src + cd.src_off;
            lim[h] = far_limit(a, cd, a.chunk_base + c);
            u32 b = (a.chunk_base + c) & ((1u << ZL_FAR_SECTION_LOG) - 1);  // chunks of this section in front of ours
            if (b > ZL_FAR_GROUP - 1) b = ZL_FAR_GROUP - 1;
            if (b > c) b = c;  // (columns exist for the launch's own chunks only)
            back[h] = b;
            list[h] = a.lists + (size_t)c * ZL_FAR_LIST_WORDS;
            count[h] = b ? list[h][0] : 0u;
        }
    }
    if (back[0] == 0 && back[1] == 0) return;
#if ZL_FAR_GATE
    // ---- is there anything to find?  (round 5)  Random bytes, small integers, i.i.d. symbols: no sampled position of such a chunk has
    // a twin further back, so none ever stops looking, and the chunk's whole list was read and all fifteen columns streamed for
    // nothing -- those chunks, not the text, were half of what this kernel did on the bench mix.  One listed position in 256 (chosen
    // by its tag: by content; the build kernel sets their keys aside) looks its bucket up in every column DIRECTLY in memory first: a
    // thousand scattered words per chunk instead of a MiB of columns and 128 KiB of list.  A chunk that finds fewer than two tags
    // there has at most a per cent of its positions matched further back and is left alone.
    {
        if (t < 2) pre_hits[t] = 0;
        __syncthreads();
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
            if (!back[h] || !count[h]) continue;
            const u32* const sparse = list[h] + 16 + 2 * (size_t)ZL_FAR_LIST_CAP;
            const u32 ns = sparse[0];
            if (ns > ZL_FAR_SPARSE_CAP) { if (t == 0) pre_hits[h] = 2; continue; }  // (long runs: every position is sampled alike)
            u32 hits = 0;
            for (u32 item = t; item < ns * back[h]; item += FAR_PROBE_THREADS) {
                const u32 kk = sparse[1 + item / back[h]], k = 1 + item % back[h];
                const u32 w = a.table[(size_t)(c0 + h - k) * FAR_COL + (kk & (FAR_COL - 1))];
                hits += (w != 0 && (w & 0x7FFFu) == (kk >> ZL_FAR_BUCKET_LOG)) ? 1u : 0u;
            }
            if (hits) atomicAdd(&pre_hits[h], hits);
        }
        __syncthreads();
#pragma unroll
        for (u32 h = 0; h < 2; h++) if (pre_hits[h] < 2) { count[h] = 0; back[h] = 0; }
        if (back[0] == 0 && back[1] == 0) return;  // (uniform; the hint words stay as the build kernel left them: empty)
    }
#endif
    // the columns the pair wants: from the nearest one (c0 itself for chunk c0 + 1, else c0 - 1) down to the farthest
    const u32 j_hi = back[1] ? c0 : c0 - 1;
    u32 j_lo = j_hi;
    if (back[0] && c0 - back[0] < j_lo) j_lo = c0 - back[0];
    if (back[1] && c0 + 1 - back[1] < j_lo) j_lo = c0 + 1 - back[1];
    const u32 most = count[0] > count[1] ? count[0] : count[1];
    for (u32 seg = 0; seg < most; seg += FAR_PROBE_THREADS * FAR_PER_THREAD) {
        // key[h][e]: bucket | tag << 14 while the entry is looking; bit 31 | (offset - position) once it has found -- a value whose
        // tag field no column word can equal, so that the lookup below needs no mask of who is still looking (bit 31 alone: no entry)
        u32 key[2][FAR_PER_THREAD];
        u32 nopen[2] = {0, 0};  // entries of this thread that are still looking
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
#pragma unroll
            for (u32 e = 0; e < FAR_PER_THREAD; e++) {
                const u32 i = seg + e * FAR_PROBE_THREADS + t;  // (consecutive threads read consecutive pairs)
                key[h][e] = 0x80000000u;
                if (i < count[h]) { key[h][e] = list[h][16 + 2 * (size_t)i]; nopen[h]++; }
            }
        }
        // ---- the columns, nearest first: an entry stops at the first one that carries its tag
        auto stream_column = [&](u32 j, u32 b) {
            // straight from memory into LDS (global_load_lds_dwordx4: no registers in between, every load of the workgroup in
            // flight at once; a wave's 64 x 16 bytes land in lane order, which is the column's own order)
            const uint4* in = (const uint4*)(a.table + (size_t)j * FAR_COL);
            uint4* out = (uint4*)colbuf[b];
#pragma unroll
            for (u32 i = 0; i < FAR_COL / 4 / FAR_PROBE_THREADS; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + t + FAR_PROBE_THREADS * i), (__attribute__((address_space(3))) void*)(out + t + FAR_PROBE_THREADS * i), 16, 0, 0);
        };
        __syncthreads();  // (the last segment's lookups are done with buffer 0)
        if (t < ZL_FAR_GROUP + 2) still_open[t] = 0;
        stream_column(j_hi, 0);
        u32 cur = 0;
        for (u32 j = j_hi;; j--) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // column j has landed in colbuf[cur]; nobody reads the other buffer any more
            // (text: every listed position finds its tag in the nearest column or two -- once no wave has an entry left looking,
            // the columns further back are neither streamed nor searched)
            if (j != j_hi && still_open[j_hi - j - 1] == 0) break;
            if (j > j_lo) stream_column(j - 1, cur ^ 1);
            const u32* col = colbuf[cur];
#pragma unroll
            for (u32 h = 0; h < 2; h++) {
                const u32 k = c0 + h - j;  // how many chunks in front of chunk c0 + h this column lies (uniform)
                if (k < 1 || k > back[h]) continue;
                if (__ballot(nopen[h] != 0) == 0) continue;  // (this wave's entries of the chunk have all found)
                u32 x[FAR_PER_THREAD];
#pragma unroll
                for (u32 e = 0; e < FAR_PER_THREAD; e++) x[e] = col[key[h][e] & (FAR_COL - 1)];  // (found entries read a harmless word)
                const u32 base = 0x80000000u | (k * ZL_CHUNK);
#pragma unroll
                for (u32 e = 0; e < FAR_PER_THREAD; e++) {
                    const bool hit = x[e] != 0 && (x[e] & 0x7FFFu) == (key[h][e] >> ZL_FAR_BUCKET_LOG);
                    if (hit) { key[h][e] = base - (x[e] >> 15); nopen[h]--; }
                }
            }
            const bool wave_open = __ballot((nopen[0] | nopen[1]) != 0) != 0;  // (all lanes vote; one writes)
            if (wave_open && (t & 63) == 0) still_open[j_hi - j] = 1;
            cur ^= 1;
            if (j == j_lo) break;
        }
        u32 found[2] = {0, 0};
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
#pragma unroll
            for (u32 e = 0; e < FAR_PER_THREAD; e++) {
                if (key[h][e] > 0x80000000u) found[h] |= 1u << e;
                key[h][e] &= 0x7FFFFFFFu;  // (entries still looking are not read again)
            }
        }
        // ---- check the hits against the bytes, measure them (up to 32 bytes) and raise the hint word of their neighbourhood: the
        // longest hit wins (ties: the larger offset).  ALL 32 bytes of both sides are asked for at once (two hits at a time):
        // one trip to memory per pair of hits instead of up to four dependent ones per hit (the sources lie up to 2 MiB back).
        // (the hint words are raised in LDS -- the column buffers are free now -- and go out to memory once, whole, instead of a
        // global atomicMax per hit, 16 000 of them per chunk of text scattered over 32 KiB: text 1.59 -> 1.17 ms per 256 MiB)
        static_assert(ZL_FAR_HINT_WORDS <= FAR_COL, "a chunk's hint words fit a column buffer");
        __syncthreads();
#pragma unroll
        for (u32 h = 0; h < 2; h++)
            for (u32 i = t; i < ZL_FAR_HINT_WORDS; i += FAR_PROBE_THREADS) colbuf[h][i] = 0;
        __syncthreads();
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
            u32* hints = colbuf[h];
#pragma unroll
            for (u32 e0 = 0; e0 < FAR_PER_THREAD; e0 += 2) {
                if (!((found[h] >> e0) & 3)) continue;
                u32 ps[2];
#pragma unroll
                for (u32 j = 0; j < 2; j++) {
                    const bool on = (found[h] >> (e0 + j)) & 1;
                    ps[j] = on ? list[h][16 + 2 * (size_t)(seg + (e0 + j) * FAR_PROBE_THREADS + t) + 1] : 0u;
                }
                uint4 a0[2], a1[2], b0[2], b1[2];
                bool wide[2];
#pragma unroll
                for (u32 j = 0; j < 2; j++) {
                    const bool on = (found[h] >> (e0 + j)) & 1;
                    const u8* pp = src[h] + ps[j];
                    const u8* qq = pp - (key[h][e0 + j] + ps[j]);
                    wide[j] = on && ps[j] + 32 <= lim[h] + 7;  // (lim + 7 bytes are readable, see far_limit)
                    a0[j] = a1[j] = b0[j] = b1[j] = make_uint4(0, 0, 0, 0);
                    b0[j].x = 1;  // (an entry that is off compares unequal)
                    if (wide[j]) { a0[j] = ld128(pp); a1[j] = ld128(pp + 16); b0[j] = ld128(qq); b1[j] = ld128(qq + 16); }
                    else if (on) { const u64 va = ld64(pp), vb = ld64(qq); a0[j].x = (u32)va; a0[j].y = (u32)(va >> 32); b0[j].x = (u32)vb; b0[j].y = (u32)(vb >> 32); }
                }
#pragma unroll
                for (u32 j = 0; j < 2; j++) {
                    if (a0[j].x != b0[j].x || a0[j].y != b0[j].y) continue;
                    const u32 p = ps[j], off = key[h][e0 + j] + ps[j];
                    u32 len = 8;
                    if (wide[j]) {  // the same steps as the loop below, from registers
                        const u64 wa[3] = {(u64)a0[j].z | ((u64)a0[j].w << 32), (u64)a1[j].x | ((u64)a1[j].y << 32), (u64)a1[j].z | ((u64)a1[j].w << 32)};
                        const u64 wb[3] = {(u64)b0[j].z | ((u64)b0[j].w << 32), (u64)b1[j].x | ((u64)b1[j].y << 32), (u64)b1[j].z | ((u64)b1[j].w << 32)};
#pragma unroll
                        for (u32 k = 0; k < 3; k++) {
                            if (len == 8 * (k + 1) && p + len < lim[h]) len += common_bytes(wa[k], wb[k]);
                        }
                    } else {
                        const u8* q = src[h] + p - off;
                        while (len < 32 && p + len < lim[h]) {
                            const u32 l = common_bytes(ld64(src[h] + p + len), ld64(q + len));
                            len += l;
                            if (l < 8) break;
                        }
                    }
                    atomicMax(hints + (p >> ZL_FAR_HINT_LOG), (len << 26) | off);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
            if (c0 + h >= nchunks) continue;
            u32* g = a.hints + (size_t)(c0 + h) * ZL_FAR_HINT_WORDS;  // (zeroed by the build kernel; this workgroup alone writes them)
            for (u32 i = t; i < ZL_FAR_HINT_WORDS; i += FAR_PROBE_THREADS) {
                const u32 v = colbuf[h][i];
                if (v && (seg == 0 || v > g[i])) g[i] = v;
            }
        }
    }
}

extern "C" hipError_t zl_launch_far_build(const zl_far_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(zl_far_build_kernel, dim3(nchunks), dim3(FAR_BUILD_THREADS), 0, stream, *a, nchunks);
    return hipGetLastError();
}
extern "C" hipError_t zl_launch_far_probe(const zl_far_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    static_assert((FAR_COL / 4) % FAR_PROBE_THREADS == 0, "a column is a whole number of 16-byte loads per thread");
    const u32 round_to = FAR_XCD_RUN ? FAR_XCDS * FAR_XCD_RUN : 1u;
    const u32 pairs = (nchunks + 1) / 2;  // a workgroup per two chunks
    hipLaunchKernelGGL(zl_far_probe_kernel, dim3(((pairs + round_to - 1) / round_to) * round_to), dim3(FAR_PROBE_THREADS), 0, stream, *a, nchunks);
    return hipGetLastError();
}

extern "C" hipError_t zl_launch_match(const zl_match_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    // levels >= 5: the larger tables
    if (a->hi && !a->dict_len) {
        if (a->far_hints) hipLaunchKernelGGL((zl_match_auto_kernel<false, false, true, true, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
        else if (a->intra) hipLaunchKernelGGL((zl_match_auto_kernel<false, false, false, true, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
        else hipLaunchKernelGGL((zl_match_auto_kernel<false, false, false, false, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
        return hipGetLastError();
    }
#ifdef ZL_TRACE
    if (a->far_hints && a->stats) { hipLaunchKernelGGL((zl_match_auto_kernel<false, true, true, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks); return hipGetLastError(); }
#endif
    if (a->far_hints) hipLaunchKernelGGL((zl_match_auto_kernel<false, false, true, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    else if (a->dict_len && a->intra) hipLaunchKernelGGL((zl_match_auto_kernel<true, false, false, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    else if (a->dict_len) hipLaunchKernelGGL((zl_match_auto_kernel<true, false, false, false>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    else if (a->intra) hipLaunchKernelGGL((zl_match_auto_kernel<false, false, false, true>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    else if (a->stats) hipLaunchKernelGGL((zl_match_auto_kernel<false, true, false, false>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    else hipLaunchKernelGGL((zl_match_auto_kernel<false, false, false, false>), dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    return hipGetLastError();
}

// =======================================================================================
// E2: entropy stage
// =======================================================================================
#define HUF_MAX_BITS 11  // zstd.c:15452 LitHufLog
#define SEQ_TILE 256

__constant__ u32 e_LL_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                  20, 22, 24, 28, 32, 40, 48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000,
                                  0x2000, 0x4000, 0x8000, 0x10000};
__constant__ u8 e_LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ u32 e_ML_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                  21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41,
                                  43, 47, 51, 59, 67, 83, 99, 0x83, 0x103, 0x203, 0x403, 0x803, 0x1003, 0x2003,
                                  0x4003, 0x8003, 0x10003};
__constant__ u8 e_ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ short e_LL_default[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ short e_ML_default[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
__constant__ short e_OF_default[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

__device__ __forceinline__ u32 hb32(u32 v) { return 31 - __clz(v); }  // v > 0

// inverse of the literal-length / match-length baseline tables (zstd.c:19533 ZSTD_LLcode,
// :19550 ZSTD_MLcode compute the same mapping from lookup tables)
__device__ __forceinline__ u32 ll_code(u32 ll) {
    if (ll < 16) return ll;
    if (ll >= 64) return hb32(ll) + 19;
    if (ll < 24) return 16 + ((ll - 16) >> 1);
    if (ll < 32) return 20 + ((ll - 24) >> 2);
    if (ll < 48) return 22 + ((ll - 32) >> 3);
    return 24;
}
__device__ __forceinline__ u32 ml_code(u32 ml) {  // ml >= 3
    const u32 mb = ml - 3;
    if (mb < 32) return mb;
    if (mb >= 128) return hb32(mb) + 36;
    if (mb < 40) return 32 + ((mb - 32) >> 1);
    if (mb < 48) return 36 + ((mb - 40) >> 2);
    if (mb < 64) return 38 + ((mb - 48) >> 3);
    if (mb < 96) return 40 + ((mb - 64) >> 4);
    return 42;
}

struct FseCTable {   // encoding side of one FSE table (zstd.c:15917 FSE_buildCTable_wksp)
    u16 state[512];  // next-state table, values in [size, 2*size)
    i32 dnb[56];     // deltaNbBits per symbol
    i32 dfs[56];     // deltaFindState per symbol
    u32 log;         // 0 for RLE mode
    int max_sym;     // last symbol described by the table header
};

// Encoding side of a dictionary's entropy section (zstd.c:27450 ZSTD_loadCEntropy), derived on the
// device from the decoder's digest of the same dictionary (zl_dict_tables): Huffman code lengths and
// codes, the three FSE encoding tables, and which symbols each of them can encode.
struct zl_cdict {
    u8 hlen[256];
    u16 hcode[256];
    u32 huf_log;
    u32 huf_ok;
    FseCTable ct[3];     // T_LL, T_OF, T_ML order
    u64 sym_mask[3];     // bit s: symbol s has a non-zero probability
    u32 ok;
    u32 pad;
};
static_assert(sizeof(zl_cdict) <= ZL_CDICT_BYTES, "ZL_CDICT_BYTES (zl_encode.h) must cover zl_cdict");

struct EntShared {
    // Two 4 KiB regions are shared by arrays with disjoint lifetimes (LDS per workgroup decides how
    // many chunks a CU works on at once): the literal histograms die when the Huffman decision is
    // made, the per-tile sequence arrays are born afterwards; the Huffman tree nodes die before the
    // first stream is coded, the bit-staging words are (re)zeroed right then.  Both regions are dead
    // when the XXH64 tile (first 8 KiB) is used at the very end.
    u32 hist[4][256];
    union {
        u32 stage[1024];  // bit staging: sequences use [0,640), Huffman wave w uses [256w, 256w+256)
        struct {
            u32 node_w[512];
            u16 node_parent[512];
            u8 node_depth[512];
        };
    };
    u8 hlen[256];
    u16 hcode[256];
    u16 sorted[256];
    u8 weights[256];
    short norm[64];
    u8 tsym[512];
    short tnorm[3][64];  // per-table normalised counts (kept for the header writer)
    u8 ttsym[3][512];    // per-table spread scratch
    FseCTable ct[3];  // K_LL, K_OF, K_ML order below
    u16 wstate[64];   // Huffman-weight FSE
    i32 wdnb[16], wdfs[16];
    u32 seq_hist[3][64];
    u32 scan_a[4], scan_b[4];
    u64 xxh_acc[4];
    u32 blk_seq[ZL_ENC_BLOCKS + 1];  // first sequence of block j (entry j+1: one past its last)
    u32 blk_lit[ZL_ENC_BLOCKS + 1];  // first literal of block j
    u32 sbits[ZL_ENC_BLOCKS][4];     // Huffman payload bits of stream k of block j
    u8 tree_buf[136];                // the tree description, copied into the first block that codes literals
    u16 cumul[4][64];  // fse_build_ctable scratch: the three sequence tables and the Huffman-weight table
    u32 small[16];     // short serial counters (code-length / weight histograms)
    u32 final_state[ZL_ENC_BLOCKS][3];  // FSE states after the last (= first in order) sequence of each block
    // control
    u32 lit_from_dict;  // the literals are coded with the dictionary's Huffman table (treeless from the first block)
    u32 lit_mode;  // 0 raw, 1 rle, 2 huffman
    u32 huf_log, huf_present, huf_maxsym, tree_size;
    u32 lit_sec_size;
    u32 seq_mode[3];
    u32 seq_hdr_size;
    u32 wpos;  // write cursor inside the frame
    u32 fail;
};
enum { T_LL = 0, T_OF = 1, T_ML = 2 };
static_assert(offsetof(EntShared, hlen) >= 8192, "the XXH64 tile aliases the two shared 4 KiB regions");
static_assert(offsetof(EntShared, hlen) == 8192 && sizeof(EntShared) - 8192 <= ZL_ENT_HANDOVER && sizeof(EntShared) % 16 == 0, "the hand-over between the two halves of the entropy stage covers everything behind the scratch regions");

// serial LSB-first bit writer used by the table-description writers (one thread)
struct BitW {
    u8* p;
    u32 pos;   // bytes written
    u64 acc;
    u32 nb;
};
__device__ __forceinline__ void bw_init(BitW& w, u8* p) { w.p = p; w.pos = 0; w.acc = 0; w.nb = 0; }
__device__ __forceinline__ void bw_add(BitW& w, u32 v, u32 n) {
    w.acc |= (u64)(v & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1))) << w.nb;
    w.nb += n;
    while (w.nb >= 8) { w.p[w.pos++] = (u8)w.acc; w.acc >>= 8; w.nb -= 8; }
}
__device__ __forceinline__ u32 bw_finish(BitW& w) {
    if (w.nb) { w.p[w.pos++] = (u8)w.acc; w.acc = 0; w.nb = 0; }
    return w.pos;
}

// Normalise counts to sum 2^log (own rounding scheme; any distribution that sums to the
// table size is valid -- cf. zstd.c:16314 FSE_normalizeCount).  Symbols rarer than one slot
// get the "less than one" marker -1.  Requires >= 2 present symbols.  One thread.
__device__ __noinline__ void fse_normalize(short* norm, const u32* count, u32 total, int max_sym, int log) {
    const u32 size = 1u << log;
    int largest = 0;
    u32 largest_c = 0, used = 0;
    for (int s = 0; s <= max_sym; s++) {
        const u32 c = count[s];
        if (c == 0) { norm[s] = 0; continue; }
        u64 scaled = ((u64)c << log);
        u32 n = (u32)(scaled / total);
        const u32 rem = (u32)(scaled - (u64)n * total);
        if (n == 0) { norm[s] = -1; used += 1; }
        else {
            if (rem * 2 >= total && n < 8) n++;  // round small probabilities to nearest
            norm[s] = (short)n;
            used += n;
        }
        if (c > largest_c) { largest_c = c; largest = s; }
    }
    if (used <= size) {
        if (used < size) norm[largest] = (short)(norm[largest] < 0 ? 1 + (size - used) : norm[largest] + (size - used));
    } else {
        u32 excess = used - size;
        while (excess) {  // take from the biggest cells first, never below one slot
            int big = -1;
            for (int s = 0; s <= max_sym; s++)
                if (norm[s] > 1 && (big < 0 || norm[s] > norm[big])) big = s;
            u32 take = (u32)(norm[big] - 1);
            if (take > excess) take = excess;
            const u32 soft = (u32)norm[big] / 4 + 1;  // spread the correction over several symbols
            if (take > soft) take = soft;
            norm[big] = (short)(norm[big] - take);
            excess -= take;
        }
    }
}

// zstd.c:16179 FSE_writeNCount: exact inverse of read_ncount in zl_decode.hip.  One thread.
__device__ __noinline__ u32 fse_write_ncount(u8* dst, const short* norm, int max_sym, int log) {
    BitW w;
    bw_init(w, dst);
    bw_add(w, (u32)(log - 5), 4);
    int remaining = (1 << log) + 1, threshold = 1 << log, nbits = log + 1;
    int sym = 0;
    bool prev_zero = false;
    while (sym <= max_sym && remaining > 1) {
        if (prev_zero) {
            int start = sym;
            while (sym <= max_sym && norm[sym] == 0) sym++;
            int run = sym - start;
            while (run >= 3) { bw_add(w, 3, 2); run -= 3; }
            bw_add(w, (u32)run, 2);
            if (sym > max_sym) break;
        }
        int count = norm[sym++];
        const int max = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        count++;
        if (count >= threshold) count += max;
        bw_add(w, (u32)count, (u32)(nbits - (count < max ? 1 : 0)));
        prev_zero = (count == 1);
        while (remaining < threshold && threshold > 1) { nbits--; threshold >>= 1; }
    }
    return bw_finish(w);
}

// zstd.c:15917 FSE_buildCTable_wksp: same spread as the decoder, then rank the cells of each
// symbol.  One thread.  tsym: LDS scratch of table size.
__device__ __noinline__ void fse_build_ctable(u16* state, i32* dnb, i32* dfs, u8* tsym, u16* cumul /* 64 entries of LDS */, const short* norm, int max_sym, int log) {
    const int size = 1 << log;
    int high = size - 1;
    cumul[0] = 0;
    for (int s = 0; s <= max_sym; s++) {
        const int n = norm[s];
        cumul[s + 1] = (u16)(cumul[s] + (n == -1 ? 1 : n));
        if (n == -1) tsym[high--] = (u8)s;
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s <= max_sym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            tsym[pos] = (u8)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    for (int u = 0; u < size; u++) {
        const int s = tsym[u];
        state[cumul[s]++] = (u16)(size + u);
    }
    int total = 0;
    for (int s = 0; s <= max_sym; s++) {
        const int n = norm[s];
        if (n == 0) { dnb[s] = ((log + 1) << 16) - size; dfs[s] = 0; }
        else if (n == -1 || n == 1) { dnb[s] = (log << 16) - size; dfs[s] = total - 1; total++; }
        else {
            const int max_bits_out = log - (int)hb32((u32)(n - 1));
            const int min_state_plus = n << max_bits_out;
            dnb[s] = (max_bits_out << 16) - min_state_plus;
            dfs[s] = total - n;
            total += n;
        }
    }
}

// ---- the same two steps by ALL 64 LANES OF ONE WAVE, lane = symbol (at most 53 symbols): the results are those of the one-thread
// versions above, cell for cell (which the Huffman-weight table still uses); what changes is the time -- a chunk's three sequence
// tables used to cost 300 000 cycles of one lane each (a division and several dependent LDS round trips per symbol and per cell),
// while the other 63 lanes of their waves waited.
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const u32 o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ u32 wave_excl_scan_u32(u32 v, u32 lane) {
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
    return inc - v;
}
#define ZL_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")  // one wave's LDS accesses execute in program order: this keeps the compiler to it
// fse_normalize by one wave.  count[s] < 2^16 and log <= 9, so the scaled count fits 32 bits.
__device__ __noinline__ void fse_normalize_wave(short* norm, const u32* count, u32 total, int max_sym, int log, u32 lane) {
    const u32 size = 1u << log;
    const bool on = (int)lane <= max_sym;
    const u32 c = on ? count[lane] : 0u;
    int n = 0;
    u32 used_l = 0;
    if (c) {
        const u32 scaled = c << log;
        u32 q = scaled / total;
        const u32 rem = scaled - q * total;
        if (q == 0) { n = -1; used_l = 1; }
        else { if (rem * 2 >= total && q < 8) q++; n = (int)q; used_l = q; }  // round small probabilities to nearest
    }
    const u32 used = wave_sum_u32(used_l);
    const int largest = 63 - (int)(wave_max_u32(c ? ((c << 6) | (63u - lane)) : 0u) & 63u);  // the largest count; the first symbol of several
    if (used <= size) {
        if (used < size && (int)lane == largest) n = n < 0 ? (int)(1 + (size - used)) : n + (int)(size - used);
    } else {
        u32 excess = used - size;  // uniform
        while (excess) {  // take from the biggest cells first, never below one slot
            const u32 b = wave_max_u32(n > 1 ? (((u32)n << 6) | (63u - lane)) : 0u);
            const int big = 63 - (int)(b & 63u);
            const u32 nb = b >> 6;
            u32 take = nb - 1;
            if (take > excess) take = excess;
            const u32 soft = nb / 4 + 1;  // spread the correction over several symbols
            if (take > soft) take = soft;
            if ((int)lane == big) n -= (int)take;
            excess -= take;
        }
    }
    if (on) norm[lane] = (short)n;
    ZL_WAVE_FENCE();
}
// fse_build_ctable by one wave.  symk: scratch of at least the table's size in bytes.
__device__ __noinline__ void fse_build_ctable_wave(u16* state, i32* dnb, i32* dfs, u8* tsym, u8* symk, const short* norm, int max_sym, int log, u32 lane) {
    const int size = 1 << log;
    const bool on = (int)lane <= max_sym;
    const int n = on ? (int)norm[lane] : 0;
    const u32 cnt = n == -1 ? 1u : (n > 0 ? (u32)n : 0u), pos_cnt = n > 0 ? (u32)n : 0u;
    const u32 cum = wave_excl_scan_u32(cnt, lane), pcum = wave_excl_scan_u32(pos_cnt, lane);
    // symbols rarer than one slot take the top cells, in symbol order from the top down
    const u64 low = __ballot(n == -1);
    const int high = size - 1 - (int)__popcll(low);
    if (n == -1) tsym[size - 1 - (int)__popcll(low & ((1ull << lane) - 1))] = (u8)lane;
    for (u32 i = 0; i < pos_cnt; i++) symk[pcum + i] = (u8)lane;  // whose cell the k-th spread cell is
    ZL_WAVE_FENCE();
    // the spread visits (j * step) & mask for j = 0, 1, ... and takes the positions at or below `high` in that order
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    const u32 per = size >= 64 ? (u32)size / 64u : 1u, j0 = lane * per;
    u32 mine = 0;
    if ((int)j0 < size) for (u32 i = 0; i < per; i++) mine += ((int)(((j0 + i) * (u32)step) & (u32)mask) <= high) ? 1u : 0u;
    u32 k = wave_excl_scan_u32(mine, lane);
    if ((int)j0 < size)
        for (u32 i = 0; i < per; i++) {
            const int p = (int)(((j0 + i) * (u32)step) & (u32)mask);
            if (p <= high) tsym[p] = symk[k++];
        }
    ZL_WAVE_FENCE();
    // a symbol's states: its cells in increasing table position (every lane walks the table; the reads are broadcasts)
    if (cnt) { u32 r = cum; for (int u = 0; u < size; u++) if (tsym[u] == (u8)lane) state[r++] = (u16)(size + u); }
    if (on) {
        if (n == 0) { dnb[lane] = ((log + 1) << 16) - size; dfs[lane] = 0; }
        else if (n == -1 || n == 1) { dnb[lane] = (log << 16) - size; dfs[lane] = (i32)cum - 1; }
        else {
            const int max_bits_out = log - (int)hb32((u32)(n - 1));
            dnb[lane] = (max_bits_out << 16) - (n << max_bits_out);
            dfs[lane] = (i32)cum - n;
        }
    }
    ZL_WAVE_FENCE();
}

// FSE encoder primitives (zstd.c:2060-2110 FSE_initCState2 / FSE_encodeSymbol)
__device__ __forceinline__ u32 fse_init_state(const u16* state, i32 dnb, i32 dfs) {
    const u32 nb = (u32)(dnb + (1 << 15)) >> 16;
    const i32 value = (i32)(nb << 16) - dnb;
    return state[(value >> nb) + dfs];
}
__device__ __forceinline__ u32 fse_encode(const u16* state, i32 dnb, i32 dfs, u32& st) {
    const u32 nb = (u32)((i32)st + dnb) >> 16;
    const u32 out = (st & ((1u << nb) - 1)) | (nb << 16);
    st = state[(i32)(st >> nb) + dfs];
    return out;  // [15:0] bits, [23:16] count
}

// OR `n` (<= 64) bits into a zeroed LDS word array at bit position `pos`.
__device__ __forceinline__ void stage_or(u32* stage, u32 pos, u64 v, u32 n) {
    if (n == 0) return;
    const u32 w = pos >> 5, sh = pos & 31;
    atomicOr(&stage[w], (u32)(v << sh));
    if (sh + n > 32) {
        const u64 hi = v >> (32 - sh);
        atomicOr(&stage[w + 1], (u32)hi);
        if (sh + n > 64) atomicOr(&stage[w + 2], (u32)(hi >> 32));
    }
}

// ---- Huffman code construction (one thread; counts in S.hist[0], symbols pre-sorted) -----
// Optimal tree by the two-queue method, depths limited to HUF_MAX_BITS by moving leaves down
// until the Kraft sum is exactly one (cf. zstd.c:17200 HUF_buildCTable_wksp / HUF_setMaxHeight
// which reach the same goal differently).  Returns table log (max length used).
__device__ __noinline__ u32 huf_build_lengths(EntShared& S, u32 m) {
    for (u32 i = 0; i < m; i++) S.node_w[i] = S.hist[0][S.sorted[i]];
    u32 q1 = 0, q2 = m, next = m;
    for (u32 k = 0; k + 1 < m; k++) {
        u32 a, b;
        if (q1 < m && (q2 >= next || S.node_w[q1] <= S.node_w[q2])) a = q1++; else a = q2++;
        if (q1 < m && (q2 >= next || S.node_w[q1] <= S.node_w[q2])) b = q1++; else b = q2++;
        S.node_w[next] = S.node_w[a] + S.node_w[b];
        S.node_parent[a] = (u16)next;
        S.node_parent[b] = (u16)next;
        next++;
    }
    const u32 root = next - 1;
    S.node_depth[root] = 0;
    u32* const bl = S.small;  // HUF_MAX_BITS + 2 counters (a dynamically indexed local array would live in scratch memory)
    for (u32 i = 0; i <= HUF_MAX_BITS + 1; i++) bl[i] = 0;
    u32 kraft = 0;  // in units of 2^-HUF_MAX_BITS
    for (int i = (int)root - 1; i >= 0; i--) {
        u32 d = S.node_depth[S.node_parent[i]] + 1u;
        if (d > HUF_MAX_BITS) d = HUF_MAX_BITS;
        S.node_depth[i] = (u8)d;
        if ((u32)i < m) { bl[d]++; kraft += 1u << (HUF_MAX_BITS - d); }
    }
    while (kraft > (1u << HUF_MAX_BITS)) {
        u32 bits = HUF_MAX_BITS - 1;
        while (bl[bits] == 0) bits--;
        bl[bits]--;
        bl[bits + 1] += 2;
        bl[HUF_MAX_BITS]--;
        kraft--;
    }
    // least frequent symbols take the longest codes
    u32 i = 0, maxlen = 0;
    for (u32 bits = HUF_MAX_BITS; bits >= 1; bits--) {
        for (u32 k = 0; k < bl[bits]; k++) S.hlen[S.sorted[i++]] = (u8)bits;
        if (bl[bits] && bits > maxlen) maxlen = bits;
    }
    return maxlen;
}

// The same construction by one WAVE (wave 3 of the entropy kernel), lane 0 for what is a chain by nature and all 64 lanes for the
// rest; the lengths are those of huf_build_lengths above.  Where the one-lane version spent its time, per phase: the two-queue
// merge re-read both queue heads from LDS several times per step (here: both heads and their successors sit in registers, one
// LDS round trip per step refills them); depths by walking every node's parent chain in index order, two dependent reads per node
// (here: level by level, all nodes at once, eleven rounds -- deeper nodes are clamped to eleven anyway); the code lengths handed
// out symbol by symbol (here: every rank looks its length up in the per-length counts).
__device__ __noinline__ u32 huf_build_lengths_wave(EntShared& S, u32 m, u32 lane) {
    for (u32 i = lane; i < m; i += 64) S.node_w[i] = S.hist[0][S.sorted[i]];
    ZL_WAVE_FENCE();
    const u32 root = 2 * m - 2;
    if (lane == 0) {
        const u32 INF = 0xFFFFFFFFu;
        u32 q1 = 0, q2 = m, next = m;
        u32 a0 = S.node_w[0], a1 = m > 1 ? S.node_w[1] : INF, b0 = INF, b1 = INF;
        for (u32 k = 0; k + 1 < m; k++) {
            u32 x, y, xi, yi;
            if (a0 <= b0) { x = a0; xi = q1++; a0 = a1; } else { x = b0; xi = q2++; b0 = b1; }
            if (a0 <= b0) { y = a0; yi = q1++; } else { y = b0; yi = q2++; }
            S.node_w[next] = x + y;
            S.node_parent[xi] = (u16)next;
            S.node_parent[yi] = (u16)next;
            next++;
            a0 = q1 < m ? S.node_w[q1] : INF;          // (four independent reads behind the write: one round trip per step)
            a1 = q1 + 1 < m ? S.node_w[q1 + 1] : INF;
            b0 = q2 < next ? S.node_w[q2] : INF;
            b1 = q2 + 1 < next ? S.node_w[q2 + 1] : INF;
        }
    }
    ZL_WAVE_FENCE();
    // depths, level by level from the root; what is deeper than HUF_MAX_BITS is clamped to it
    for (u32 i = lane; i < root; i += 64) S.node_depth[i] = 0xFF;
    if (lane == 0) S.node_depth[root] = 0;
    ZL_WAVE_FENCE();
    for (u32 r = 1; r <= HUF_MAX_BITS; r++) {
        for (u32 i = lane; i < root; i += 64)
            if (S.node_depth[i] == 0xFF && S.node_depth[S.node_parent[i]] == r - 1) S.node_depth[i] = (u8)r;
        ZL_WAVE_FENCE();
    }
    u32* const bl = S.small;  // HUF_MAX_BITS + 2 counters
    if (lane <= HUF_MAX_BITS + 1) bl[lane] = 0;
    ZL_WAVE_FENCE();
    u32 kraft_l = 0;
    for (u32 i = lane; i < m; i += 64) {
        u32 d = S.node_depth[i];
        if (d > HUF_MAX_BITS) d = HUF_MAX_BITS;
        atomicAdd(&bl[d], 1u);
        kraft_l += 1u << (HUF_MAX_BITS - d);
    }
    u32 kraft = wave_sum_u32(kraft_l);  // in units of 2^-HUF_MAX_BITS
    ZL_WAVE_FENCE();
    if (lane == 0) {
        while (kraft > (1u << HUF_MAX_BITS)) {
            u32 bits = HUF_MAX_BITS - 1;
            while (bl[bits] == 0) bits--;
            bl[bits]--;
            bl[bits + 1] += 2;
            bl[HUF_MAX_BITS]--;
            kraft--;
        }
    }
    ZL_WAVE_FENCE();
    // least frequent symbols take the longest codes: rank i (ascending count) gets the longest length whose counts reach past i
    u32 cnt_b[HUF_MAX_BITS + 1];
#pragma unroll
    for (u32 b = 1; b <= HUF_MAX_BITS; b++) cnt_b[b] = bl[b];
    u32 maxlen = 0;
#pragma unroll
    for (u32 b = 1; b <= HUF_MAX_BITS; b++) if (cnt_b[b]) maxlen = b;
    for (u32 i = lane; i < m; i += 64) {
        u32 acc = 0, len = 1;
#pragma unroll
        for (u32 b = HUF_MAX_BITS; b >= 1; b--) { if (i >= acc && i < acc + cnt_b[b]) len = b; acc += cnt_b[b]; }
        S.hlen[S.sorted[i]] = (u8)len;
    }
    ZL_WAVE_FENCE();
    return maxlen;
}

// Huffman tree description (zstd.c:16907 HUF_writeCTable_wksp): weights of symbols
// 0..maxsym-1, FSE-compressed when that is smaller (zstd.c:16840 HUF_compressWeights),
// else 4 bits each.  By one WAVE (wave 3 of the entropy kernel): the weights, their histogram, the normalised counts and the
// encoding table on all lanes (lane = symbol / weight; the same cells as the one-thread functions above), the bit stream of
// the weights -- a chain by nature -- on lane 0.  A full chunk hides this behind its FSE state chains; a batch of small
// objects has nothing to hide it behind, and one thread took 173 000 cycles per object for it.
// Returns bytes written (the same in all lanes), 0 if the table cannot be described.
__device__ __noinline__ u32 huf_write_tree_wave(EntShared& S, u8* dst, u32 lane) {
    const u32 maxsym = S.huf_maxsym, log = S.huf_log;
    u32* const wc = S.small;
    if (lane < 16) wc[lane] = 0;
    ZL_WAVE_FENCE();
    u32 maxw_l = 0;
    for (u32 sy = lane; sy < maxsym; sy += 64) {
        const u32 w = S.hlen[sy] ? log + 1 - S.hlen[sy] : 0u;
        S.weights[sy] = (u8)w;
        atomicAdd(&wc[w], 1u);
        maxw_l = w > maxw_l ? w : maxw_l;
    }
    ZL_WAVE_FENCE();
    const u32 maxw = wave_max_u32(maxw_l);
    const u32 maxcount = wave_max_u32(lane <= maxw ? wc[lane] : 0u);
    u32 fse_size = 0;
    if (maxsym > 1 && maxcount != maxsym && maxcount != 1) {  // (uniform)
        int wlog = (int)hb32(maxsym - 1) - 1;  // a little below log2(#weights)
        if (wlog > 6) wlog = 6;
        if (wlog < 5) wlog = 5;
        fse_normalize_wave(S.norm, wc, maxsym, (int)maxw, wlog, lane);
        ZL_WAVE_FENCE();
        fse_build_ctable_wave(S.wstate, S.wdnb, S.wdfs, S.tsym, (u8*)S.cumul[3], S.norm, (int)maxw, wlog, lane);
        ZL_WAVE_FENCE();
        if (lane == 0) {
            u8* p = dst + 1;
            const u32 hs = fse_write_ncount(p, S.norm, (int)maxw, wlog);
            // two interleaved states, last weight first (zstd.c:16090 FSE_compress_usingCTable_generic)
            BitW w;
            bw_init(w, p + hs);
            int ip = (int)maxsym;
            u32 s1, s2;
            if (maxsym & 1) {
                ip--; s1 = fse_init_state(S.wstate, S.wdnb[S.weights[ip]], S.wdfs[S.weights[ip]]);
                ip--; s2 = fse_init_state(S.wstate, S.wdnb[S.weights[ip]], S.wdfs[S.weights[ip]]);
                ip--; { const u32 c = S.weights[ip]; const u32 o = fse_encode(S.wstate, S.wdnb[c], S.wdfs[c], s1); bw_add(w, o & 0xFFFF, o >> 16); }
            } else {
                ip--; s2 = fse_init_state(S.wstate, S.wdnb[S.weights[ip]], S.wdfs[S.weights[ip]]);
                ip--; s1 = fse_init_state(S.wstate, S.wdnb[S.weights[ip]], S.wdfs[S.weights[ip]]);
            }
            while (ip > 0) {
                ip--; { const u32 c = S.weights[ip]; const u32 o = fse_encode(S.wstate, S.wdnb[c], S.wdfs[c], s2); bw_add(w, o & 0xFFFF, o >> 16); }
                ip--; { const u32 c = S.weights[ip]; const u32 o = fse_encode(S.wstate, S.wdnb[c], S.wdfs[c], s1); bw_add(w, o & 0xFFFF, o >> 16); }
            }
            bw_add(w, s2, (u32)wlog);
            bw_add(w, s1, (u32)wlog);
            bw_add(w, 1, 1);
            fse_size = hs + bw_finish(w);
        }
        fse_size = (u32)__shfl((int)fse_size, 0, 64);
    }
    if (fse_size > 1 && fse_size < maxsym / 2 && fse_size < 128) {
        if (lane == 0) dst[0] = (u8)fse_size;
        ZL_WAVE_FENCE();
        return 1 + fse_size;
    }
    if (maxsym > 128) return 0;
    if (lane == 0) dst[0] = (u8)(127 + maxsym);
    {   // (at most 128 weights: a byte of two per lane)
        const u32 sy = 2 * lane;
        if (sy < maxsym) {
            const u32 hi = S.weights[sy], lo = (sy + 1 < maxsym) ? S.weights[sy + 1] : 0;
            dst[1 + lane] = (u8)((hi << 4) | lo);
        }
    }
    ZL_WAVE_FENCE();
    return 1 + (maxsym + 1) / 2;
}

// One backward Huffman stream by one wave (zstd.c:17690 HUF_compress1X_usingCTable_internal):
// the last symbol is written first.  Tiles of 512 symbols; each lane codes 8 consecutive
// symbols, a wave scan places its bits, LDS atomicOr assembles words, whole bytes are flushed.
__device__ u32 huf_encode_stream(const u8* hlen /* LDS */, const u16* hcode /* LDS */, u32* stage /* 256 words, zeroed */, const u8* lit, u32 cnt, u8* out) {
    const u32 lane = threadIdx.x & 63;
    u32 written = 0, carry_bits = 0;
    // the 8 symbols of a lane are 8 consecutive bytes (read back to front): one load, fetched a tile ahead
    u64 nextv = 0;
    if (cnt >= 8 * (lane + 1)) nextv = ld64(lit + cnt - 8 * (lane + 1));
    for (u32 e0 = 0; e0 < cnt; e0 += 512) {
        const u32 first = e0 + lane * 8;  // emission index of this lane's first symbol
        u64 v = nextv;
        if (first < cnt && first + 8 > cnt) {  // the ragged front of the stream: byte by byte
            v = 0;
            for (u32 j = 0; first + j < cnt; j++) v |= (u64)lit[cnt - 1 - first - j] << (56 - 8 * j);
        }
        if (first + 512 + 8 <= cnt) nextv = ld64(lit + cnt - (first + 512) - 8);
        u64 lo = 0, hi = 0;
        u32 nlo = 0, nhi = 0;
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            if (first + j < cnt) {
                const u32 sym = (u32)(v >> (56 - 8 * j)) & 0xFF;  // byte cnt-1-(first+j)
                const u32 len = hlen[sym];
                const u64 code = hcode[sym];
                if (j < 4) { lo |= code << nlo; nlo += len; }
                else { hi |= code << nhi; nhi += len; }
            }
        }
        const u32 mine = nlo + nhi;
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 x = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += x;
        }
        const u32 tile_bits = __shfl(inc, 63, 64);
        const u32 pos = carry_bits + inc - mine;
        stage_or(stage, pos, lo, nlo);
        stage_or(stage, pos + nlo, hi, nhi);
        __builtin_amdgcn_wave_barrier();
        const u32 total = carry_bits + tile_bits;
        const u32 nbytes = total >> 3;
        const u8* sb = (const u8*)stage;
        {   // whole dwords first (global memory takes unaligned dword stores), then the odd bytes
            const u32 nd = nbytes >> 2;
            for (u32 i = lane; i < nd; i += 64) { const u32 x = stage[i]; __builtin_memcpy(out + written + 4 * i, &x, 4); }
            if (lane < (nbytes & 3)) out[written + 4 * nd + lane] = sb[4 * nd + lane];
        }
        __builtin_amdgcn_wave_barrier();
        // keep the partial byte, clear the rest
        const u32 keep = (total & 7) ? sb[nbytes] : 0;
        __builtin_amdgcn_wave_barrier();
        for (u32 i = lane; i < 256; i += 64) stage[i] = (i == 0) ? keep : 0;
        __builtin_amdgcn_wave_barrier();
        written += nbytes;
        carry_bits = total & 7;
    }
    // end marker
    if (lane == 0) {
        const u32 last = (stage[0] & 0xFF) | (1u << carry_bits);
        out[written] = (u8)last;
    }
    __builtin_amdgcn_wave_barrier();
    for (u32 i = lane; i < 256; i += 64) stage[i] = 0;
    return written + 1;
}

// workgroup-wide byte copy, 16 bytes per thread and trip (global memory takes unaligned vectors)
__device__ __forceinline__ void copy_wg(u8* dst, const u8* src, u32 n) {
    const u32 t = threadIdx.x, body = n & ~15u;
    for (u32 i = t * 16; i < body; i += ENC_THREADS * 16) {
        const uint4 v = ld128(src + i);
        __builtin_memcpy(dst + i, &v, 16);
    }
    if (t < n - body) dst[body + t] = src[body + t];
}

__device__ __forceinline__ u32 min_gain(u32 n) { return (n >> 6) + 2; }  // zstd.c:19607 ZSTD_minGain (strategy < btultra)

__device__ __forceinline__ void wr_le(u8* p, u64 v, u32 nbytes) {
    for (u32 i = 0; i < nbytes; i++) p[i] = (u8)(v >> (8 * i));
}

// zstd.c:27089 ZSTD_writeFrameHeader for a single-segment frame of n bytes.
__device__ __noinline__ u32 write_frame_header(u8* p, u32 n, u32 checksum, u32 dict_id) {
    wr_le(p, ZL_MAGIC, 4);
    const u32 fcs_flag = n < 256 ? 0 : (n < 65536 + 256 ? 1 : 2);
    const u32 dflag = dict_id == 0 ? 0 : (dict_id < 256 ? 1 : (dict_id < 65536 ? 2 : 3));
    p[4] = (u8)((fcs_flag << 6) | (1u << 5) | (checksum ? 4 : 0) | dflag);
    u32 pos = 5;
    const u32 dbytes = dflag == 3 ? 4 : dflag;
    wr_le(p + pos, dict_id, dbytes);
    pos += dbytes;
    if (fcs_flag == 0) { p[pos++] = (u8)n; }
    else if (fcs_flag == 1) { wr_le(p + pos, n - 256, 2); pos += 2; }
    else { wr_le(p + pos, n, 4); pos += 4; }
    return pos;
}

// zstd.c:27089 ZSTD_writeFrameHeader for the single-frame layout: a frame of `total` bytes whose blocks are written by
// many chunks.  Single_Segment when the content fits the window (zstd.c:27100), else a Window_Descriptor of 2^wlog.
__device__ __forceinline__ u32 dict_id_bytes(u32 dict_id) { return dict_id == 0 ? 0u : (dict_id < 256 ? 1u : (dict_id < 65536 ? 2u : 4u)); }
__device__ __forceinline__ u32 joined_header_size(u64 total, u32 wlog, u32 dict_id) {
    const bool single = total <= (1ull << wlog);
    const u32 fcs = total < 256 ? (single ? 1u : 2u) : (total < 65536 + 256 ? 2u : (total < 0xFFFFFFFFull ? 4u : 8u));
    return 5 + (single ? 0u : 1u) + dict_id_bytes(dict_id) + fcs;
}
__device__ __noinline__ u32 write_frame_header_joined(u8* p, u64 total, u32 wlog, u32 checksum, u32 dict_id) {
    wr_le(p, ZL_MAGIC, 4);
    const bool single = total <= (1ull << wlog);
    const u32 dbytes = dict_id_bytes(dict_id);
    const u32 fcs = joined_header_size(total, wlog, dict_id) - 5 - (single ? 0u : 1u) - dbytes;
    const u32 fcs_flag = fcs == 1 ? 0u : (fcs == 2 ? 1u : (fcs == 4 ? 2u : 3u));
    p[4] = (u8)((fcs_flag << 6) | (single ? (1u << 5) : 0u) | (checksum ? 4u : 0u) | (dbytes == 4 ? 3u : dbytes));
    u32 pos = 5;
    if (!single) p[pos++] = (u8)((wlog - 10) << 3);
    wr_le(p + pos, dict_id, dbytes);
    pos += dbytes;
    wr_le(p + pos, fcs == 2 ? total - 256 : total, fcs);
    return pos + fcs;
}

// Bits (x256) a symbol costs in an FSE table, read off its encoding transform: between min and min + 1
// bits, by the share of its states that take the larger count (the estimate zstd.c:15560 FSE_bitCost makes)
__device__ __forceinline__ u32 fse_bit_cost(int dnb, u32 log) {
    const u32 min_nb = (u32)dnb >> 16;
    const u32 threshold = (min_nb + 1) << 16;
    const u32 delta = threshold - ((u32)dnb + (1u << log));
    return ((min_nb + 1) << 8) - ((delta << 8) >> log);
}

// Dictionary entropy for the encoder: one workgroup turns the decoder's digest into encoding tables.
__global__ __launch_bounds__(ENC_THREADS) void zl_cdict_build_kernel(const zl_dict_tables* dt, zl_cdict* out) {
    __shared__ u8 hlen[256];
    __shared__ u8 tsym[512];
    __shared__ u16 cumul[64];
    const u32 t = threadIdx.x;
    hlen[t] = 0;
    __syncthreads();
    const u32 hlog = dt->huf_log;
    if (t == 0) { out->ok = (dt->status == 0 && hlog >= 1 && hlog <= 12) ? 1u : 0u; out->huf_log = hlog; out->huf_ok = 1; }
    // Huffman: code length per symbol from the decode table, codes in the decoder's cell order
    for (u32 i = t; i < (1u << hlog) && hlog <= 12; i += ENC_THREADS) { const u32 e = dt->huf[i]; hlen[e & 0xFF] = (u8)(e >> 8); }
    __syncthreads();
    {
        const u32 len = hlen[t];
        u32 code = 0;
        if (len) {
            const u32 w = hlog + 1 - len;
            u32 start = 0;
            for (u32 q = 0; q < 256; q++) {
                const u32 lq = hlen[q];
                if (!lq) continue;
                const u32 wq = hlog + 1 - lq;
                if (wq < w || (wq == w && q < t)) start += 1u << (wq - 1);
            }
            code = start >> (w - 1);
        }
        out->hlen[t] = (u8)len;
        out->hcode[t] = (u16)code;
    }
    // FSE: symbol of every cell from the decode cell, then the encoder's view of the same table
    for (int k = 0; k < 3; k++) {
        __syncthreads();
        const u32 log = k == T_LL ? dt->ll_log : (k == T_ML ? dt->ml_log : dt->of_log);
        const u32 size = 1u << log;
        const u64* cells = k == T_LL ? dt->fse_ll : (k == T_ML ? dt->fse_ml : dt->fse_of);
        const int nsym = k == T_LL ? 36 : (k == T_ML ? 53 : 32);
        for (u32 u = t; u < size; u += ENC_THREADS) {
            const u64 cell = cells[u];
            const u32 base = (u32)cell, bits = (u32)(cell >> 48) & 0xFF;
            u32 sym = 0;
            if (k == T_OF) sym = bits;
            else for (int q = 0; q < nsym; q++) { const u32 b = k == T_LL ? e_LL_base[q] : e_ML_base[q]; if (b == base) sym = (u32)q; }
            tsym[u] = (u8)sym;
        }
        __syncthreads();
        if (t == 0) {
            FseCTable& ct = out->ct[k];
            u32 cnt[64];
            for (int q = 0; q < 64; q++) cnt[q] = 0;
            for (u32 u = 0; u < size; u++) cnt[tsym[u]]++;
            u64 mask = 0;
            int max_sym = 0;
            u32 run = 0;
            for (int q = 0; q < nsym; q++) { cumul[q] = (u16)run; run += cnt[q]; if (cnt[q]) { mask |= 1ull << q; max_sym = q; } }
            for (u32 u = 0; u < size; u++) { const u32 q = tsym[u]; ct.state[cumul[q]++] = (u16)(size + u); }
            int total = 0;
            for (int q = 0; q < nsym; q++) {
                const int n = (int)cnt[q];
                if (n == 0) { ct.dnb[q] = (int)((log + 1) << 16) - (int)size; ct.dfs[q] = 0; }
                else if (n == 1) { ct.dnb[q] = (int)(log << 16) - (int)size; ct.dfs[q] = total - 1; total++; }
                else {
                    const int max_bits_out = (int)log - (int)hb32((u32)(n - 1));
                    const int min_state_plus = n << max_bits_out;
                    ct.dnb[q] = (max_bits_out << 16) - min_state_plus;
                    ct.dfs[q] = total - n;
                    total += n;
                }
            }
            ct.log = log;
            ct.max_sym = max_sym;
            out->sym_mask[k] = mask;
        }
    }
}

// position of block boundary `j` of `nbk`: the byte target j*n/nbk moved to the end of the match
// that covers it, if one does.  Returns {first sequence, first literal} of the block behind it.
__device__ __noinline__ void block_boundary(const zl_seq* seqs, u32 nseq, u32 nlit, u32 n, u32 j, u32 nbk, u32* first_seq, u32* first_lit) {
    if (j == 0) { *first_seq = 0; *first_lit = 0; return; }
    if (j >= nbk) { *first_seq = nseq; *first_lit = nlit; return; }
    const u32 target = (u32)(((u64)j * n) / nbk);
    u32 lo = 0, hi = nseq;  // first sequence whose match ends behind the target
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        const zl_seq s = seqs[mid];
        if (s.pos + s.match_len > target) hi = mid; else lo = mid + 1;
    }
    if (lo == nseq) { *first_seq = nseq; *first_lit = nlit - (n - target); return; }  // inside the trailing literals
    const zl_seq s = seqs[lo];
    if (target <= s.pos) { *first_seq = lo; *first_lit = s.lit_cum - (s.pos - target); }  // inside the literal run in front of match lo
    else { *first_seq = lo + 1; *first_lit = s.lit_cum; }                                  // inside match lo: the block ends with it
}

// literal length and offset code value of sequence i inside its block (zstd.c:25186 ZSTD_seqToCodes input):
// the first sequence of a block counts its literals from the block start and always spells its
// offset out, so that no block depends on the history its predecessor leaves behind.
__device__ __forceinline__ void seq_fields(const zl_seq* seqs, u32 i, u32 blk_first_seq, u32 blk_first_lit, const zl_seq& s, u32& ll, u32& ob, bool exact) {
    const bool first = i == blk_first_seq;
    const u32 prev_cum = first ? blk_first_lit : seqs[i - 1].lit_cum;
    const u32 prev_off = first ? 0u : seqs[i - 1].offset;
    ll = s.lit_cum - prev_cum;
    ob = exact ? s.pos : ((ll > 0 && s.offset == prev_off) ? 1u : s.offset + 3);  // (exact: exact_rep_codes has left the code value in `pos`)
}

// Levels >= 6: the offset code value of every sequence of ONE block under the decoder's three-entry repeat-offset history
// (zstd.c:19665 ZSTD_storeSeq / zstd.c:44291 ZSTD_updateRep: with literals in front, the most recent three offsets are codes
// 1, 2, 3; without, the second and third and "the first minus one"), followed sequence by sequence on one lane and written
// into the records' `pos` field (nothing reads positions once the block boundaries are known).  What the history holds when
// the block starts is what the block in front left behind, which this lane does not know: an unknown entry is 0, which no
// offset equals, and every code it emits says what the decoder's history looks like afterwards whatever lay underneath -- a new
// offset pushes, a repeat code moves its entry to the front.  Eight records are fetched per trip: the chain depends on the
// history, not on memory.
__device__ __noinline__ void exact_rep_codes(zl_seq* seqs, u32 first, u32 end, u32 first_lit) {
    u32 r0 = 0, r1 = 0, r2 = 0, prev_cum = first_lit;
    for (u32 base = first; base < end; base += 8) {
        uint4 rec[8];
#pragma unroll
        for (int k = 0; k < 8; k++) if (base + k < end) __builtin_memcpy(&rec[k], &seqs[base + k], 16);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (base + k >= end) break;
            const u32 o = rec[k].z, ll = rec[k].x - prev_cum;  // {lit_cum, match_len, offset, pos}
            prev_cum = rec[k].x;
            u32 ob;
            if (ll) {
                if (o == r0) ob = 1;
                else if (o == r1) { ob = 2; r1 = r0; r0 = o; }
                else if (o == r2) { ob = 3; r2 = r1; r1 = r0; r0 = o; }
                else { ob = o + 3; r2 = r1; r1 = r0; r0 = o; }
            } else {
                if (o == r1) { ob = 1; r1 = r0; r0 = o; }
                else if (o == r2) { ob = 2; r2 = r1; r1 = r0; r0 = o; }
                else if (r0 > 1 && o == r0 - 1) { ob = 3; r2 = r1; r1 = r0; r0 = o; }
                else { ob = o + 3; r2 = r1; r1 = r0; r0 = o; }
            }
            seqs[base + k].pos = ob;
        }
    }
}

// (The serial helpers above are __noinline__: inlined, their registers weigh on a kernel that is held to 64
// of them -- measured 2.51 -> 2.39 ms.  Taking the phase stamps out at compile time as well removed most of
// the remaining spills and was SLOWER, 2.56 ms: the allocator's choices matter more than the spill count.)
#ifndef ZL_ENT_WGS
#define ZL_ENT_WGS 8   // workgroups per CU the register budget is cut for (8: 64 VGPRs, the LDS footprint's own limit)
#endif
// PHASE 0: the whole stage in one kernel.  PHASE 1 / 2: the same code as two kernels -- statistics, tables and FSE chains; then the
// blocks' bit packing and the frame -- with the workgroup's shared state (everything in EntShared behind its two scratch regions, 11 KiB)
// handed over through memory.  Each half gets a register allocation of its own: as ONE kernel the stage wants 115 vector registers and is held
// to the 64 that eight workgroups per CU allow (65 spilled, 96 scalar ones).
#ifndef ZL_ENT2_WGS
#define ZL_ENT2_WGS 4   // ... the second half's register budget: 128 (it is the half that wants them: held to 64 it spills 250)
#endif
template <int PHASE>
__global__ __launch_bounds__(ENC_THREADS, PHASE == 2 ? ZL_ENT2_WGS : ZL_ENT_WGS) void zl_entropy_kernel(zl_entropy_args a, u32 nchunks) {
    __shared__ EntShared S;
    ZL_POISON(S, 8);
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const u32 c = a.order ? a.order[zl_spread(blockIdx.x, nchunks)] : zl_spread(blockIdx.x, nchunks);
    const zl_chunk_desc cd = a.chunks[c];
    const u8* src = a.src + cd.src_off;
    const u32 n = cd.src_size;
    const zl_seq* seqs = a.seqs + (size_t)c * a.seq_stride;
    const u32 nseq = a.meta[c].nseq, nlit = a.meta[c].nlit;
    const u8* lit = nseq ? a.lits + (size_t)c * a.lit_stride : src;  // (a chunk without a sequence is all literals: the match kernels leave them where they are)
    // per-sequence scratch of the FSE chains, one array per table (ZL_SST_BYTES per sequence slot): the three codes as bytes,
    // the three states BEFORE each sequence as 16-bit words.  (Round 2 kept one 8-byte record per sequence; a chain lane then
    // needed a record per step, and its loop ran at the latency of those loads: one memory round trip per eight steps.  Now
    // eight steps' codes are ONE 8-byte load, fetched four loads ahead.)
    u8* const sst_base = (u8*)a.seq_states + (size_t)c * a.seq_stride * ZL_SST_BYTES;
    u8* const code_k[3] = {sst_base, sst_base + a.seq_stride, sst_base + 2 * (size_t)a.seq_stride};                    // T_LL, T_OF, T_ML
    unsigned short* const state_k[3] = {(unsigned short*)(sst_base + 3 * (size_t)a.seq_stride), (unsigned short*)(sst_base + 5 * (size_t)a.seq_stride),
                                        (unsigned short*)(sst_base + 7 * (size_t)a.seq_stride)};
    if (a.small_done && a.frame_size[c] != 0xFFFFFFFFu) return;  // (zl_entropy_small_kernel wrote this one's frame)
    u8* out = a.out + (size_t)c * a.out_stride;
    const u32 out_limit = a.out_stride - 16;  // what a frame may occupy while it is being built
    const bool prof = a.stats != nullptr && t == 0;
    u64 pt[6] = {0, 0, 0, 0, 0, 0};
    u64 p_last = prof ? __builtin_amdgcn_s_memtime() : 0;
#define ENT_STAMP(k_) if (prof) { const u64 now__ = __builtin_amdgcn_s_memtime(); pt[k_] += now__ - p_last; p_last = now__; }

    // ---- frame header; the blocks follow
    // (single-frame layout: only the frame's first chunk carries a header, only its last one ends the frame)
    const bool joined = a.joined != 0;
    const bool frame_first = !joined || a.chunk_base + c == 0, frame_last = !joined || a.chunk_base + c + 1 == a.total_chunks;
    const u32 fh = joined ? (frame_first ? joined_header_size(a.total_size, a.wlog, a.dict_id) : 0u)
                          : (n < 256 ? 6u : (n < 65536 + 256 ? 7u : 9u)) + dict_id_bytes(a.dict_id);
    u32 nbk = (n >> 13) < 1 ? 1u : ((n >> 13) > ZL_ENC_BLOCKS ? (u32)ZL_ENC_BLOCKS : (n >> 13));
    {
        const u32 worth = (nseq * 4u + nlit) >> 12;
        if (worth < nbk) nbk = worth ? worth : 1u;
    }
    const bool exact = a.exact_rep != 0;
    u8* const handover = a.handover ? a.handover + (size_t)c * ZL_ENT_HANDOVER : nullptr;
    if (PHASE != 2) {
    for (u32 i = t; i < 4 * 256; i += ENC_THREADS) (&S.hist[0][0])[i] = 0;
    for (u32 i = t; i < 3 * 64; i += ENC_THREADS) (&S.seq_hist[0][0])[i] = 0;
    for (u32 i = t; i < 1024; i += ENC_THREADS) S.stage[i] = 0;
    if (t == 0) {
        S.fail = 0;
        if (!joined) write_frame_header(out, n, a.checksum, a.dict_id);
        else if (frame_first) write_frame_header_joined(out, a.total_size, a.wlog, a.checksum, a.dict_id);  // (the checksum itself: one XXH64 over the whole content, appended by the host -- zl_xxh64.h)
    }
    // ---- blocks: up to ZL_ENC_BLOCKS per chunk, cut at sequence boundaries near equal byte counts.
    // They share one set of entropy tables (the first block describes them, the others use the
    // repeat modes, zstd.c:20745 / :21059), which keeps the format overhead at ~12 bytes per block
    // and lets a decoder run the blocks' Huffman streams and FSE chains side by side.
    // (... and no more blocks than the chunk's content is worth: every block costs a header, two section headers and a stream
    // split, about 20 bytes, which a chunk that is a few hundred sequences of long matches -- records that repeat, ratio 100 and
    // more -- cannot carry eight times: one block per 4 KiB of sequences (counted at 4 bytes) and literals)
    // The boundaries: for each, the first sequence whose match ends behind its byte target.  Searched WIDE: 256 threads sample the
    // sequences' ends at equal strides, every boundary then names its stride from the samples in LDS, and the threads test that
    // stride's sequences at once -- three memory round trips where a binary search per boundary made fifteen in a row while the
    // other 247 threads waited.  Same boundaries as block_boundary() finds.
    if (nbk > 1 && nseq) {
        u32* const samp = S.stage;          // [256] the sampled ends (the staging words are zeroed again below)
        u32* const lo_of = S.stage + 256;   // [nbk] per boundary: the first sequence that ends behind the target (minimum over the candidates)
        const u32 stride = (nseq + ENC_THREADS - 1) / ENC_THREADS;
        {
            const u32 i = t * stride;
            u32 e = 0xFFFFFFFFu;
            if (i < nseq) { const zl_seq q = seqs[i]; e = q.pos + q.match_len; }
            samp[t] = e;
            if (t < ZL_ENC_BLOCKS + 1) lo_of[t] = nseq;
        }
        __syncthreads();
        // candidates of boundary j: the stride behind the last sample that does not end behind the target (or the first stride)
        for (u32 item = t; item < (nbk - 1) * (stride + 1); item += ENC_THREADS) {
            const u32 j = 1 + item / (stride + 1), off = item % (stride + 1);
            const u32 target = (u32)(((u64)j * n) / nbk);
            u32 lo = 0, hi = ENC_THREADS;  // the samples are ascending: first sample that ends behind the target
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (samp[mid] > target) hi = mid; else lo = mid + 1; }
            const u32 base = lo ? (lo - 1) * stride : 0u, i = base + off;  // the answer lies in [base, base + stride]
            if (i < nseq) { const zl_seq q = seqs[i]; if (q.pos + q.match_len > target) atomicMin(&lo_of[j], i); }
        }
        __syncthreads();
        if (t <= ZL_ENC_BLOCKS) {
            u32 fs = 0, fl = 0;
            if (t >= nbk) { fs = nseq; fl = nlit; }
            else if (t > 0) {
                const u32 target = (u32)(((u64)t * n) / nbk), lo = lo_of[t];
                if (lo == nseq) { fs = nseq; fl = nlit - (n - target); }  // inside the trailing literals
                else {
                    const zl_seq q = seqs[lo];
                    if (target <= q.pos) { fs = lo; fl = q.lit_cum - (q.pos - target); }  // inside the literal run in front of match lo
                    else { fs = lo + 1; fl = q.lit_cum; }                                  // inside match lo: the block ends with it
                }
            }
            S.blk_seq[t] = fs; S.blk_lit[t] = fl;
        }
        __syncthreads();
        for (u32 i = t; i < 256 + ZL_ENC_BLOCKS + 1; i += ENC_THREADS) S.stage[i] = 0;
    } else if (t <= ZL_ENC_BLOCKS) block_boundary(seqs, nseq, nlit, n, t, nbk, &S.blk_seq[t], &S.blk_lit[t]);
    __syncthreads();
    // levels >= 6: a lane per block follows the decoder's repeat-offset history through its block (the other waves go on to the literal
    // histogram, which does not look at sequences; everybody meets again in front of the sequence statistics)
    if (exact && nseq && t < nbk) exact_rep_codes(const_cast<zl_seq*>(seqs), S.blk_seq[t], S.blk_seq[t + 1], S.blk_lit[t]);

    // =========================== statistics of the whole chunk: literal histogram, sequence codes and their histograms
    // (zstd.c:20692 ZSTD_compressLiterals, zstd.c:25265 ZSTD_buildSequencesStatistics).  The serial
    // pieces that follow run side by side on different waves: the Huffman tree next to the three FSE
    // tables, the tree description next to the FSE state chains.
    {   // 16 literals per thread and trip (one byte per trip leaves a single load in flight per thread);
        // four histogram copies, one per wave, keep the atomics apart
        // ZERO bytes are counted in a register and added once: lanes that hit one bin in one instruction are served one after the
        // other, and zero is the byte that binary data is full of (a quarter of the literals of 20-bit integers stored in 32:
        // the pass took three times as long as on random bytes)
        // Four copies of every bin, side by side, one per lane modulo four (round 2: one copy per WAVE, which keeps waves apart
        // but not the lanes of one instruction -- and those are what the LDS serialises: literals over a ten-letter alphabet put
        // six lanes on every bin): the lanes that hit one byte value spread over four neighbouring banks.
        u32* const H = &S.hist[0][0] + (lane & 3);
        const u32 body = a.raw_lits ? 0u : (nlit & ~15u);  // (negative levels store their literals: nothing to count)
        u32 zeros = 0;
        for (u32 i = t * 16; i < body; i += ENC_THREADS * 16) {
            const uint4 v = ld128(lit + i);
            const u32 w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 b0 = w4[k] & 0xFF, b1 = (w4[k] >> 8) & 0xFF, b2 = (w4[k] >> 16) & 0xFF, b3 = w4[k] >> 24;
                if (b0) atomicAdd(&H[4 * b0], 1u);
                if (b1) atomicAdd(&H[4 * b1], 1u);
                if (b2) atomicAdd(&H[4 * b2], 1u);
                if (b3) atomicAdd(&H[4 * b3], 1u);
                zeros += (b0 == 0) + (b1 == 0) + (b2 == 0) + (b3 == 0);
            }
        }
        if (zeros) atomicAdd(&H[0], zeros);
        if (!a.raw_lits && t < nlit - body) atomicAdd(&H[4 * (u32)lit[body + t]], 1u);
    }
    // (the three code histograms likewise in four copies per bin, in the staging words -- 3 KiB that nothing else uses during this
    // pass --, summed into seq_hist behind the barrier: a text chunk's 18 000 sequences share a handful of codes)
    u32* const SH = S.stage + (lane & 3);
    if (exact) zl_sync_global();  // (the code values were written by other lanes)
    if (nseq) {
        for (u32 j = 0; j < nbk; j++) {
            const u32 f_seq = S.blk_seq[j], e_seq = S.blk_seq[j + 1], f_lit = S.blk_lit[j];
            // two consecutive sequences per thread and trip: the second's predecessor is the first, so three loads -- all in flight
            // together -- serve two sequences (one sequence per trip made two dependent trips to memory for each)
            for (u32 i = f_seq + 2 * t; i < e_seq; i += 2 * ENC_THREADS) {
                const bool two = i + 1 < e_seq, first = i == f_seq;
                const zl_seq s0 = seqs[i];
                zl_seq s1 = s0, sp = s0;
                if (two) s1 = seqs[i + 1];
                if (!first) sp = seqs[i - 1];
                const u32 prev_cum = first ? f_lit : sp.lit_cum, prev_off = first ? 0u : sp.offset;  // (seq_fields: a block's first sequence spells its offset out)
                const u32 ll0 = s0.lit_cum - prev_cum, ob0 = exact ? s0.pos : ((ll0 > 0 && s0.offset == prev_off) ? 1u : s0.offset + 3);
                const u32 cl0 = ll_code(ll0), co0 = hb32(ob0), cm0 = ml_code(s0.match_len);
                atomicAdd(&SH[4 * (64 * T_LL + cl0)], 1u);
                atomicAdd(&SH[4 * (64 * T_OF + co0)], 1u);
                atomicAdd(&SH[4 * (64 * T_ML + cm0)], 1u);
                code_k[T_LL][i] = (u8)cl0; code_k[T_OF][i] = (u8)co0; code_k[T_ML][i] = (u8)cm0;
                if (two) {
                    const u32 ll1 = s1.lit_cum - s0.lit_cum, ob1 = exact ? s1.pos : ((ll1 > 0 && s1.offset == s0.offset) ? 1u : s1.offset + 3);
                    const u32 cl1 = ll_code(ll1), co1 = hb32(ob1), cm1 = ml_code(s1.match_len);
                    atomicAdd(&SH[4 * (64 * T_LL + cl1)], 1u);
                    atomicAdd(&SH[4 * (64 * T_OF + co1)], 1u);
                    atomicAdd(&SH[4 * (64 * T_ML + cm1)], 1u);
                    code_k[T_LL][i + 1] = (u8)cl1; code_k[T_OF][i + 1] = (u8)co1; code_k[T_ML][i + 1] = (u8)cm1;
                }
            }
        }
    }
    zl_sync_global();  // (the codes travel to the chain lanes through memory)
    if (t < 3 * 64) (&S.seq_hist[0][0])[t] = S.stage[4 * t] + S.stage[4 * t + 1] + S.stage[4 * t + 2] + S.stage[4 * t + 3];
    const u32 cnt_t = (&S.hist[0][0])[4 * t] + (&S.hist[0][0])[4 * t + 1] + (&S.hist[0][0])[4 * t + 2] + (&S.hist[0][0])[4 * t + 3];
    __syncthreads();
    S.hist[0][t] = cnt_t;
    __syncthreads();
    // rank sort of present symbols (ascending count, then symbol)
    {
        u32 rank = 0, present = 0;
        for (u32 s = 0; s < 256; s++) {
            const u32 cs = S.hist[0][s];
            if (cs) {
                present++;
                if (cnt_t && (cs < cnt_t || (cs == cnt_t && s < t))) rank++;
            }
        }
        if (cnt_t) S.sorted[rank] = (u16)t;
        if (t == 0) {
            S.huf_present = present;
            u32 mode = 0;
            if (nlit > 63 && present == 1) mode = 1;   // zstd.c:20725 minLitSize, :20750 RLE
            else if (nlit > 63 && present >= 2) mode = 2;
            if (a.raw_lits) mode = 0;
            S.lit_mode = mode;
            S.tree_size = 0;
        }
    }
    S.hlen[t] = 0;
    __syncthreads();
    // With a dictionary, a frame of a single block may take its entropy tables from it: treeless
    // literals (zstd.c:20745: minLitSize drops to 6 when a table can be repeated) and FSE repeat
    // modes, both from the very first block -- no table descriptions at all, which is where small
    // inputs spend their bytes.
    const zl_cdict* dtab = (const zl_cdict*)a.cdict;
    const bool dict_tables = dtab != nullptr && nbk == 1 && dtab->ok;
    if (t == 0) S.lit_from_dict = 0;
    ENT_STAMP(0)
    // ---- wave 3: the Huffman tree  |  waves 0..2: one FSE table each
    if (wave == 3) {  // (uniform inside the wave: every lane reads the same words)
        bool go = S.lit_mode == 2;
        // zstd.c:18076 (HUF_compress_internal): "heuristic: probably not compressible enough"
        if (go && S.hist[0][S.sorted[S.huf_present - 1]] <= (nlit >> 7) + 4) { go = false; if (lane == 0) S.lit_mode = 0; }
        if (go) {
            const u32 log = huf_build_lengths_wave(S, S.huf_present, lane);
            const u32 hc = S.hist[0][4 * lane] | S.hist[0][4 * lane + 1] | S.hist[0][4 * lane + 2] | S.hist[0][4 * lane + 3];
            const u64 any = __ballot(hc != 0);  // the largest symbol present: in the four of the highest lane that holds one
            const u32 top = 63u - (u32)__builtin_clzll(any);
            u32 maxsym = 4 * top + 3;
            while (S.hist[0][maxsym] == 0) maxsym--;
            if (lane == 0) { S.huf_log = log; S.huf_maxsym = maxsym; }
        }
    }
    // table choice + construction (zstd.c:21012 ZSTD_selectEncodingType, strategy < lazy branch :21034-21059): waves 0..2 build one
    // table each, all 64 lanes at it (lane = symbol)
    if (nseq && wave < 3) {
        const int k = (int)wave;
        const int max_code = k == T_LL ? 35 : (k == T_ML ? 52 : 31);
        const int def_log = k == T_OF ? 5 : 6, max_log = k == T_OF ? 8 : 9;
        const u32* h = S.seq_hist[k];
        const u32 hc = (int)lane <= max_code ? h[lane] : 0u;
        const u64 present = __ballot(hc != 0);
        const int max_sym = present ? 63 - (int)__builtin_clzll(present) : 0;
        const u32 most = wave_max_u32(hc);
        FseCTable& ct = S.ct[k];
        short* norm = S.tnorm[k];
        u8* tsym = S.ttsym[k];
        u8* symk = (u8*)S.hist[1 + k];  // (the per-wave literal histograms are dead since their sum went into hist[0])
        u32 mode;
        if (most == nseq) mode = (nseq <= 2) ? 0 : 1;
        else if (nseq < ((1u << def_log) * 8u >> 3) || most < (nseq >> (def_log - 1))) mode = 0;
        else mode = 2;
        if (k == T_OF && mode == 0 && max_sym > 28) mode = 2;  // default offset table stops at code 28
        if (lane == 0) ct.max_sym = max_sym;
        if (mode == 0) {
            const short* def = k == T_LL ? e_LL_default : (k == T_ML ? e_ML_default : e_OF_default);
            const int dmax = k == T_LL ? 35 : (k == T_ML ? 52 : 28);
            if ((int)lane <= dmax) norm[lane] = def[lane];
            ZL_WAVE_FENCE();
            fse_build_ctable_wave(ct.state, ct.dnb, ct.dfs, tsym, symk, norm, dmax, def_log, lane);
            if (lane == 0) ct.log = (u32)def_log;
        } else if (mode == 1) {
            if (lane == 0) ct.log = 0;
        } else {
            int log = (int)hb32(nseq - 1) - 2;  // zstd.c:16220 FSE_optimalTableLog idea
            if (log > max_log) log = max_log;
            const int min_log = (int)hb32((u32)max_sym) + 2;
            if (log < min_log) log = min_log;
            if (log < 5) log = 5;
            if (log > max_log) log = max_log;
            fse_normalize_wave(norm, h, nseq, max_sym, log, lane);
            fse_build_ctable_wave(ct.state, ct.dnb, ct.dfs, tsym, symk, norm, max_sym, log, lane);
            if (lane == 0) ct.log = (u32)log;
        }
        ZL_WAVE_FENCE();
        if (dict_tables && mode != 1) {
            // Repeat_Mode against the table just built (zstd.c:21060-21110 weighs the same three costs):
            // every code that occurs must have a cell in the dictionary's table, and the bits it
            // spends there must not exceed ours plus our table's description.  All 64 lanes at it (lane = symbol):
            // a batch of small objects with a dictionary spends its time here otherwise.
            const FseCTable& dct = dtab->ct[k];
            if ((present & ~dtab->sym_mask[k]) == 0) {
                const u32 dlog = dct.log;
                const bool mine = hc != 0;  // (present: lanes <= max_code <= 52 < 56)
                const u32 cost_dict = wave_sum_u32(mine ? hc * fse_bit_cost(dct.dnb[lane], dlog) : 0u);
                u32 cost_own = wave_sum_u32(mine ? hc * fse_bit_cost(ct.dnb[lane], ct.log) : 0u);
                if (mode == 2) cost_own += 256u * 8u * (((u32)max_sym + 1) * 5 / 8 + 1);
                if (cost_dict <= cost_own) {
                    mode = 3;
                    for (u32 q = lane; q < (1u << dlog); q += 64) ct.state[q] = dct.state[q];
                    if (lane < 56) { ct.dnb[lane] = dct.dnb[lane]; ct.dfs[lane] = dct.dfs[lane]; }
                    if (lane == 0) { ct.log = dlog; ct.max_sym = dct.max_sym; }
                }
            }
        }
        if (lane == 0) S.seq_mode[k] = mode;
    }
    __syncthreads();
    ENT_STAMP(1)
    if (dict_tables && nlit >= 6) {
        // bits with the dictionary's codes against bits with our own codes plus our tree description
        const u32 h = S.hist[0][t], dl = dtab->hlen[t];
        u32 v_dict = h * dl, v_own = h * S.hlen[t], bad = (h && !dl) ? 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { v_dict += __shfl_xor(v_dict, d, 64); v_own += __shfl_xor(v_own, d, 64); bad += __shfl_xor(bad, d, 64); }
        if (lane == 0) { S.scan_a[wave] = v_dict; S.scan_b[wave] = v_own | (bad ? 0x80000000u : 0u); }
        __syncthreads();
        if (t == 0) {
            const u32 bits_dict = S.scan_a[0] + S.scan_a[1] + S.scan_a[2] + S.scan_a[3];
            const u32 own_raw = (S.scan_b[0] & 0x7FFFFFFFu) + (S.scan_b[1] & 0x7FFFFFFFu) + (S.scan_b[2] & 0x7FFFFFFFu) + (S.scan_b[3] & 0x7FFFFFFFu);
            const bool invalid = ((S.scan_b[0] | S.scan_b[1] | S.scan_b[2] | S.scan_b[3]) & 0x80000000u) != 0;
            // our own table costs its description: about half a byte per symbol up to the largest one present
            const u32 bits_own = S.lit_mode == 2 ? own_raw + 8 * (S.huf_maxsym / 2 + 2) : 8 * nlit;
            if (!invalid && S.lit_mode != 1 && bits_dict < bits_own) { S.lit_from_dict = 1; S.lit_mode = 2; S.huf_log = dtab->huf_log; }
        }
        __syncthreads();
        if (S.lit_from_dict) { S.hlen[t] = dtab->hlen[t]; S.hcode[t] = dtab->hcode[t]; }
        __syncthreads();
    }
    if (S.lit_mode == 2) {
        // codes: same cell order as the decoder's table (ascending weight, ascending symbol)
        if (!S.lit_from_dict) {
            const u32 len = S.hlen[t];
            if (len) {
                const u32 log = S.huf_log, w = log + 1 - len;
                u32 start = 0;
                for (u32 s = 0; s < 256; s++) {
                    const u32 ls = S.hlen[s];
                    if (!ls) continue;
                    const u32 ws = log + 1 - ls;
                    if (ws < w || (ws == w && s < t)) start += 1u << (ws - 1);
                }
                S.hcode[t] = (u16)(start >> (w - 1));
            }
        }
        // payload bits of every (block, stream): a wave per stream, code lengths looked up per literal
        for (u32 item = wave; item < nbk * 4; item += 4) {
            const u32 j = item >> 2, k = item & 3;
            const u32 l0 = S.blk_lit[j], nl = S.blk_lit[j + 1] - l0;
            const bool four = nl >= 256;  // zstd.c:20716 singleStream = srcSize < 256
            const u32 seg = (nl + 3) / 4;
            u32 s0 = 0, s1 = 0;
            if (four) { s0 = k * seg; s1 = (k == 3) ? nl : s0 + seg; }
            else if (k == 0) s1 = nl;
            u32 bits = 0;
            const u8* sp = lit + l0 + s0;
            const u32 cnt = s1 - s0, body = cnt & ~7u;
            for (u32 i = lane * 8; i < body; i += 64 * 8) {
                const u64 v = ld64(sp + i);
#pragma unroll
                for (int q = 0; q < 8; q++) bits += S.hlen[(u32)(v >> (8 * q)) & 0xFF];
            }
            if (lane < cnt - body) bits += S.hlen[sp[body + lane]];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) bits += __shfl_xor(bits, d, 64);
            if (lane == 0) S.sbits[j][k] = bits;
        }
    }
    __syncthreads();
    ENT_STAMP(2)
    // ---- wave 3: the tree description and the literal decision  |  wave 0: the FSE state chains
    if (wave == 3 && S.lit_mode == 2) {  // (uniform inside the wave)
        const bool from_dict = S.lit_from_dict != 0;
        const u32 tree = from_dict ? 0u : huf_write_tree_wave(S, S.tree_buf, lane);
        if (lane == 0) {
            u32 total = 0;
            for (u32 j = 0; j < nbk; j++) {
                const bool four = S.blk_lit[j + 1] - S.blk_lit[j] >= 256;
                for (u32 k = 0; k < (four ? 4u : 1u); k++) total += (S.sbits[j][k] + 8) >> 3;  // payload + end marker
                total += four ? 6 : 0;
            }
            S.tree_size = tree;
            if ((!from_dict && tree == 0) || tree + total >= nlit - min_gain(nlit)) S.lit_mode = 0;  // zstd.c:20770: not worth it -> raw
        }
    }
    // The FSE state chains (zstd.c:21163-21222) are serial per table and per block -- and the blocks
    // are independent, so the 3 x nbk chains run side by side on the lanes of one wave.  Each lane
    // leaves the state *before* every sequence in the scratch record; the bits that state emits are
    // cut from it when the bitstream is assembled.
    if (nseq && wave == 0) {
        const u32 j = lane / 3, k = lane % 3;
        const bool on = lane < 3 * nbk;
        const u32 e_seq = on ? S.blk_seq[j + 1] : 0u, ns = on ? e_seq - S.blk_seq[j] : 0u;
        u32 max_ns = ns;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const u32 o = __shfl_xor(max_ns, d, 64); max_ns = o > max_ns ? o : max_ns; }
        const FseCTable& ct = S.ct[k];
        const u32 log = ct.log;
        const u8* const codes = code_k[k];
        unsigned short* const states = state_k[k];
        u32 st = 0;
        // Step e of a chain codes sequence e_seq - 1 - e; the eight steps of a trip read the eight code bytes at
        // [e_seq - 8 - e0, e_seq - e0): one load.  Four trips' loads are in flight: the chain itself only touches LDS, and the loop
        // used to wait for memory once per trip.
        auto fetch = [&](u32 e0) -> u64 {  // (bytes below the block's first sequence belong to its neighbour: read, never used)
            if (e0 >= ns) return 0;
            const int b = (int)e_seq - 8 - (int)e0;
            return b >= 0 ? ld64(codes + b) : (ld64(codes) << (8 * (u32)(-b)));  // (the array's first bytes, moved up to where a full trip would have them)
        };
        u64 pre0 = fetch(0), pre1 = fetch(8), pre2 = fetch(16), pre3 = fetch(24);
        for (u32 e0 = 0; e0 < max_ns; e0 += 8) {
            const u64 cur = pre0;
            pre0 = pre1; pre1 = pre2; pre2 = pre3; pre3 = fetch(e0 + 32);
            u32 before[8];
            i32 dn[8], df[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {  // the eight symbols' transforms first: sixteen LDS reads that do not depend on the state
                const u32 sym = (u32)(cur >> (8 * (7 - q))) & 0xFFu;
                dn[q] = ct.dnb[sym]; df[q] = ct.dfs[sym];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {  // then the eight steps, each one dependent table read
                const u32 e = e0 + q;
                before[q] = 0;
                if (e < ns && log) {
                    if (e == 0) st = fse_init_state(ct.state, dn[q], df[q]);
                    else { before[q] = st; (void)fse_encode(ct.state, dn[q], df[q], st); }
                }
            }
            if (e0 + 8 <= ns && e_seq >= e0 + 8) {  // a full trip: eight states, one 16-byte store (step q's sequence is the (7 - q)-th of them)
                uint4 w;
                w.x = before[7] | (before[6] << 16); w.y = before[5] | (before[4] << 16); w.z = before[3] | (before[2] << 16); w.w = before[1] | (before[0] << 16);
                __builtin_memcpy(states + (e_seq - 8 - e0), &w, 16);
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) if (e0 + q < ns) states[e_seq - 1 - (e0 + q)] = (unsigned short)before[q];
            }
        }
        if (on) S.final_state[j][k] = st;
    }
    zl_sync_global();  // the saved states travel to the assembling threads through memory
    }  // (PHASE != 2)
    if (PHASE == 1) {  // hand the shared state over: everything behind the two scratch regions
        const uint4* from = (const uint4*)((const u8*)&S + 8192);
        uint4* to = (uint4*)handover;
        for (u32 i = t; i < (sizeof(EntShared) - 8192 + 15) / 16; i += ENC_THREADS) to[i] = from[i];
        if (prof) for (int k = 0; k < 6; k++) atomicAdd((unsigned long long*)&a.stats[k], (unsigned long long)pt[k]);
        return;
    }
    if (PHASE == 2) {
        const uint4* from = (const uint4*)handover;
        uint4* to = (uint4*)((u8*)&S + 8192);
        for (u32 i = t; i < (sizeof(EntShared) - 8192 + 15) / 16; i += ENC_THREADS) to[i] = from[i];
    }
    // the tree nodes shared this region: staging words must be zero before bits are OR-ed in
    for (u32 i = t; i < 1024; i += ENC_THREADS) S.stage[i] = 0;
    __syncthreads();
    ENT_STAMP(5)

    // =========================== the blocks, one after the other (zstd.c:26833 ZSTD_compressBlock_internal)
    u32 cur = fh;             // write cursor inside the frame
    bool fail = false;        // uniform: the frame outgrew its staging slot -> raw block
    bool tree_out = S.lit_from_dict != 0;  // uniform: a block has carried the Huffman tree (or the dictionary supplies the table)
    bool tables_out = false;  // uniform: a block has carried the sequence tables
    u32 last_hdr = 0, last_hdr_val = 0;  // where the last block header went and what it holds (for the Last_Block bit)
    const u32 tree_sz = S.tree_size;
    for (u32 j = 0; j < nbk && !fail; j++) {
        const u32 f_seq = S.blk_seq[j], e_seq = S.blk_seq[j + 1], f_lit = S.blk_lit[j];
        const u32 ns = e_seq - f_seq, nl = S.blk_lit[j + 1] - f_lit;
        // regenerated size of the block: its literals plus its matches
        if (ns == 0 && nl == 0) continue;
        u8* blk = out + cur + 3;
        const u8* blit = lit + f_lit;
        // ---- literals section
        const bool four = nl >= 256;
        const u32 lhs_c = 3 + (nl >= 1024) + (nl >= 16384);       // compressed-literals header size
        const u32 lhs_r = 1 + (nl >= 32) + (nl >= 4096);          // raw / rle header size
        u32 mode = 0, csize = 0, tree = 0;
        if (S.lit_mode == 1 && nl > 0) mode = 1;
        else if (S.lit_mode == 2 && nl > 0) {
            for (u32 k = 0; k < (four ? 4u : 1u); k++) csize += (S.sbits[j][k] + 8) >> 3;
            csize += four ? 6 : 0;
            tree = tree_out ? 0 : tree_sz;
            if (lhs_c + tree + csize < lhs_r + nl) mode = tree_out ? 3 : 2;
        }
        u32 lit_sec;
        if (mode >= 2) {
            lit_sec = lhs_c + tree + csize;
            if (cur + 3 + lit_sec > out_limit) { fail = true; break; }
            u8* p = blk + lhs_c + tree;
            if (t == 0) {
                const u32 cs = tree + csize;
                const u32 sf = four ? (lhs_c == 3 ? 1 : (lhs_c == 4 ? 2 : 3)) : 0;
                const u32 bits = lhs_c == 3 ? 10 : (lhs_c == 4 ? 14 : 18);
                const u64 h = (u64)mode | (sf << 2) | ((u64)nl << 4) | ((u64)cs << (4 + bits));
                wr_le(blk, h, lhs_c);
                if (four) {
                    wr_le(p, (S.sbits[j][0] + 8) >> 3, 2);
                    wr_le(p + 2, (S.sbits[j][1] + 8) >> 3, 2);
                    wr_le(p + 4, (S.sbits[j][2] + 8) >> 3, 2);
                }
            }
            if (tree) for (u32 i = t; i < tree; i += ENC_THREADS) blk[lhs_c + i] = S.tree_buf[i];
            if (four) {
                const u32 seg = (nl + 3) / 4;
                u32 off = 6;
                for (u32 k = 0; k < wave; k++) off += (S.sbits[j][k] + 8) >> 3;
                const u32 s0 = wave * seg, cnt = (wave == 3) ? nl - 3 * seg : seg;
                huf_encode_stream(S.hlen, S.hcode, S.stage + 256 * wave, blit + s0, cnt, p + off);
            } else if (wave == 0) {
                huf_encode_stream(S.hlen, S.hcode, S.stage, blit, nl, p);
            }
            tree_out = true;
        } else if (mode == 1) {
            lit_sec = lhs_r + 1;
            if (cur + 3 + lit_sec > out_limit) { fail = true; break; }
            if (t == 0) {
                const u64 h = 1u | ((lhs_r == 1 ? 0u : (lhs_r == 2 ? 1u : 3u)) << 2) | ((u64)nl << (lhs_r == 1 ? 3 : 4));
                wr_le(blk, h, lhs_r);
                blk[lhs_r] = blit[0];
            }
        } else {
            lit_sec = lhs_r + nl;
            if (cur + 3 + lit_sec > out_limit) { fail = true; break; }
            if (t == 0) {
                const u64 h = 0u | ((lhs_r == 1 ? 0u : (lhs_r == 2 ? 1u : 3u)) << 2) | ((u64)nl << (lhs_r == 1 ? 3 : 4));
                wr_le(blk, h, lhs_r);
            }
            copy_wg(blk + lhs_r, blit, nl);
        }
        __syncthreads();
        for (u32 i = t; i < 1024; i += ENC_THREADS) S.stage[i] = 0;
        __syncthreads();

        ENT_STAMP(3)
        // ---- sequences section (zstd.c:21146 ZSTD_encodeSequences)
        u8* sq = blk + lit_sec;
        u32 seq_sec = 0;
        if (ns == 0) {
            if (t == 0) sq[0] = 0;
            seq_sec = 1;
        } else {
            if (cur + 3 + lit_sec + 512 > out_limit) { fail = true; break; }
            if (t == 0) {
                u32 pos = 0;
                if (ns < 128) sq[pos++] = (u8)ns;
                else if (ns < 0x7F00) { sq[pos++] = (u8)((ns >> 8) + 0x80); sq[pos++] = (u8)ns; }
                else { sq[pos++] = 0xFF; wr_le(sq + pos, ns - 0x7F00, 2); pos += 2; }
                if (tables_out) sq[pos++] = 0xFC;  // Repeat_Mode x 3
                else {
                    sq[pos++] = (u8)((S.seq_mode[T_LL] << 6) | (S.seq_mode[T_OF] << 4) | (S.seq_mode[T_ML] << 2));
                    for (int k = 0; k < 3; k++) {
                        const u32 m = S.seq_mode[k];
                        if (m == 1) sq[pos++] = (u8)S.ct[k].max_sym;
                        else if (m == 2) pos += fse_write_ncount(sq + pos, S.tnorm[k], S.ct[k].max_sym, (int)S.ct[k].log);
                    }
                }
                S.seq_hdr_size = pos;
            }
            tables_out = true;
            __syncthreads();
            // tiles of 256 sequences, last sequence first
            u8* bs = sq + S.seq_hdr_size;
            const u32 bs_pos = cur + 3 + lit_sec + S.seq_hdr_size;
            u32 written = 0, carry_bits = 0;
            // (a tile's records -- the sequence, its predecessor's literal count and offset, the three saved states -- are asked for
            // one tile AHEAD: the loop used to wait for memory once per tile, 72 times per text chunk)
            struct TileRec { u32 lit_cum, match_len, offset, prev_cum, prev_off, bo, bl, bm; };
            auto fetch_tile = [&](u32 e) -> TileRec {
                TileRec r = {0, 0, 0, 0, 0, 0, 0, 0};
                if (e < ns) {
                    const u32 i = e_seq - 1 - e;
                    const zl_seq sq_ = seqs[i];
                    r.lit_cum = sq_.lit_cum; r.match_len = sq_.match_len; r.offset = exact ? sq_.pos : sq_.offset;  // (exact: the code value itself, exact_rep_codes)
                    if (i == f_seq) { r.prev_cum = f_lit; r.prev_off = 0; }  // (seq_fields: a block's first sequence counts its literals from the block start and spells its offset out)
                    else { const zl_seq pq = seqs[i - 1]; r.prev_cum = pq.lit_cum; r.prev_off = pq.offset; }
                    if (e > 0) { r.bo = state_k[T_OF][i]; r.bl = state_k[T_LL][i]; r.bm = state_k[T_ML][i]; }  // states before this sequence
                }
                return r;
            };
            TileRec nxt_rec = fetch_tile(t);
            for (u32 e0 = 0; e0 < ns; e0 += SEQ_TILE) {
                const u32 e = e0 + t;  // emission index; sequence index = e_seq-1-e
                const TileRec rec = nxt_rec;
                nxt_rec = fetch_tile(e + SEQ_TILE);
                u64 extra = 0;   // LL extra | ML extra | OF extra (in stream order)
                u32 nextra = 0;
                u32 cl = 0, cm = 0, co = 0;
                if (e < ns) {
                    const u32 ll = rec.lit_cum - rec.prev_cum;
                    const u32 ob = exact ? rec.offset : ((ll > 0 && rec.offset == rec.prev_off) ? 1u : rec.offset + 3);
                    cl = ll_code(ll); cm = ml_code(rec.match_len); co = hb32(ob);
                    const u32 nl_ = e_LL_bits[cl], nm = e_ML_bits[cm];
                    extra = (u64)(ll - e_LL_base[cl]);
                    extra |= (u64)(rec.match_len - e_ML_base[cm]) << nl_;
                    extra |= (u64)(ob - (1u << co)) << (nl_ + nm);
                    nextra = nl_ + nm + co;
                }
                // per sequence: [OF state][ML state][LL state][LL extra][ML extra][OF extra]; the first
                // sequence emitted only sets the states up
                u64 lo = 0;
                u32 nlo = 0;
                if (e < ns && e > 0) {
                    const u32 bo = rec.bo, bl = rec.bl, bm = rec.bm;
                    const u32 no = S.ct[T_OF].log ? (u32)((i32)bo + S.ct[T_OF].dnb[co]) >> 16 : 0u;
                    const u32 nm_ = S.ct[T_ML].log ? (u32)((i32)bm + S.ct[T_ML].dnb[cm]) >> 16 : 0u;
                    const u32 nl2 = S.ct[T_LL].log ? (u32)((i32)bl + S.ct[T_LL].dnb[cl]) >> 16 : 0u;
                    lo = (u64)(bo & ((1u << no) - 1));
                    nlo = no;
                    lo |= (u64)(bm & ((1u << nm_) - 1)) << nlo;
                    nlo += nm_;
                    lo |= (u64)(bl & ((1u << nl2) - 1)) << nlo;
                    nlo += nl2;
                }
                u32 ex_a, ex_b, tot_a, tot_b;
                block_scan2(S.scan_a, S.scan_b, nlo + nextra, 0u, ex_a, ex_b, tot_a, tot_b);
                const u32 pos = carry_bits + ex_a;
                stage_or(S.stage, pos, lo, nlo);
                stage_or(S.stage, pos + nlo, extra, nextra);
                __syncthreads();
                const u32 total = carry_bits + tot_a;
                const u32 nbytes = total >> 3;
                if (bs_pos + written + nbytes + 32 > out_limit) fail = true;  // uniform
                const u8* sb = (const u8*)S.stage;
                if (!fail) {  // whole dwords (global memory takes unaligned dword stores), then the odd bytes
                    const u32 nd = nbytes >> 2;
                    for (u32 i = t; i < nd; i += ENC_THREADS) { const u32 x = S.stage[i]; __builtin_memcpy(bs + written + 4 * i, &x, 4); }
                    if (t < (nbytes & 3)) bs[written + 4 * nd + t] = sb[4 * nd + t];
                }
                const u32 keep = (total & 7) ? sb[nbytes] : 0;
                __syncthreads();
                const u32 used = (total >> 5) + 3 < 1024u ? (total >> 5) + 3 : 1024u;  // (only the words this tile touched)
                for (u32 i = t; i < used; i += ENC_THREADS) S.stage[i] = (i == 0) ? keep : 0;
                __syncthreads();
                if (fail) break;
                written += nbytes;
                carry_bits = total & 7;
            }
            if (fail) break;
            // flush the three states (ML, OF, LL) and the end marker (zstd.c:21225-21233)
            if (t == 0) {
                BitW w;
                bw_init(w, bs + written);
                w.acc = S.stage[0] & 0xFF;
                w.nb = carry_bits;
                bw_add(w, S.final_state[j][T_ML], S.ct[T_ML].log);
                bw_add(w, S.final_state[j][T_OF], S.ct[T_OF].log);
                bw_add(w, S.final_state[j][T_LL], S.ct[T_LL].log);
                bw_add(w, 1, 1);
                S.wpos = S.seq_hdr_size + written + bw_finish(w);
            }
            __syncthreads();
            seq_sec = S.wpos;
            __syncthreads();
            if (t == 0) S.stage[0] = 0;
        }
        // ---- block header (zstd.c:27060): Last_Block is set on the final one below
        const u32 bsize = lit_sec + seq_sec;
        if (t == 0) wr_le(out + cur, (2u << 1) | (bsize << 3), 3);
        last_hdr = cur;
        last_hdr_val = (2u << 1) | (bsize << 3);
        cur += 3 + bsize;
        __syncthreads();
        ENT_STAMP(4)
    }

    // =========================== frame assembly (zstd.c:27060-27068, :27733)
    const bool raw = fail || n == 0 || (cur - fh - 3) + min_gain(n) >= n;  // zstd.c:25534: not enough gain -> raw block
    u32 fsize;
    if (raw) {
        u8* blk = out + fh + 3;
        copy_wg(blk, src, n);
        if (t == 0) wr_le(out + fh, (frame_last ? 1u : 0u) | (0u << 1) | (n << 3), 3);
        fsize = fh + 3 + n;
    } else {
        if (t == 0 && frame_last) wr_le(out + last_hdr, last_hdr_val | 1u, 3);  // Last_Block
        fsize = cur;
    }
    if (a.checksum && !joined) {
        __syncthreads();
        const u32 x = zl_xxh64_low32((u8*)&S, S.xxh_acc, S.scan_a, src, n);  // first 8 KiB of S are dead by now
        if (t == 0) wr_le(out + fsize, x, 4);
        fsize += 4;
    }
    if (t == 0) a.frame_size[c] = fsize;
    ENT_STAMP(5)
#undef ENT_STAMP
    if (prof) {
        for (int k = 0; k < 6; k++) atomicAdd((unsigned long long*)&a.stats[k], (unsigned long long)pt[k]);
        atomicAdd((unsigned long long*)&a.stats[6], 1ull);
    }
}


// ---------------------------------------------------------------------------------------
// zl_entropy_small_kernel (round 6): the entropy stage of a SMALL object that rides on a dictionary's tables -- config C4's
// shape: 100 000 objects of a few hundred bytes, one frame each.  zl_entropy_kernel gives every object a workgroup of 256 threads,
// zeroes and ranks four histograms, builds a Huffman tree and three FSE tables and then finds, for nearly every such object, that the
// dictionary's tables are cheaper than its own plus their descriptions: 110 000 cycles of barriers and one-lane phases per object,
// 2.3 of the 4.1 ms a call of 100 000 objects spends in kernels.  The reference does not ask either: with a dictionary whose tables
// are valid, literals of at most 1 KiB take the dictionary's Huffman table as it is (zstd.c:18020 HUF_compress_internal: preferRepeat
// below the lazy strategies) and fewer than 1000 sequences take its FSE tables in Repeat_Mode (zstd.c:21034 ZSTD_selectEncodingType).
// Here: ONE WAVE per object, four objects per workgroup, the dictionary's tables copied into LDS once per workgroup, no workgroup
// barrier behind that copy.  Eligible: a dictionary with entropy tables, no checksum, levels 3 - 4 (repeat offset 1 only), content below
// 16 KiB (one block), at most 1023 literals (one Huffman stream, a 3-byte section header), at most 127 sequences, every literal and
// every code present in the dictionary's tables.  Everything else gets frame_size = 0xFFFFFFFF and is done by zl_entropy_kernel
// behind this launch (a.small_done).
//   literals    Treeless_Literals_Block on the dictionary's table when that beats raw (from 6 literals up, zstd.c:20745), else raw
//   sequences   the bitstream of zstd.c:21146 written front to back by all lanes in step (uniform: the three states and the bit
//               container are the same in every lane; lane 0 stores), the sequence's fields handed round with lane shuffles
// ---------------------------------------------------------------------------------------
#define SMALL_G 16            // lanes per object: FOUR objects share a wave.  What an object costs is its serial chains -- a wave spends the same instructions on them whether
#define SMALL_OBJECTS 16      // 64 lanes follow or 16 -- and with a wave per object the kernel was bound by instruction issue (100 000 objects: 0.73 ms at six waves per SIMD)
#define SMALL_MAX_LIT 1023u
#define SMALL_MAX_SEQ 127u
#define SMALL_STAGE_WORDS 360 // one Huffman stream of at most 1023 codes of at most 11 bits, and its end mark
struct SmallShared {
    u16 state[3][512];
    i32 dnb[3][56], dfs[3][56];
    u32 log[3];
    u8 hlen[256];
    u16 hcode[256];
    u32 stage[SMALL_OBJECTS][SMALL_STAGE_WORDS];
};

#define SMALL_OWN_TABLE_FROM 64u  // literal bytes from which on an object whose literals stay raw under the dictionary's codes goes to the general kernel
__global__ __launch_bounds__(SMALL_G * SMALL_OBJECTS) void zl_entropy_small_kernel(zl_entropy_args a, u32 nchunks) {
    __shared__ SmallShared S;
    const u32 t = threadIdx.x, lane = t & (SMALL_G - 1), slot = t / SMALL_G;  // (lane: inside the object's group)
    const zl_cdict* const cd_ = (const zl_cdict*)a.cdict;
    for (u32 i = t; i < 3 * 512; i += SMALL_G * SMALL_OBJECTS) S.state[i >> 9][i & 511] = cd_->ct[i >> 9].state[i & 511];
    for (u32 i = t; i < 3 * 56; i += SMALL_G * SMALL_OBJECTS) { S.dnb[i / 56][i % 56] = cd_->ct[i / 56].dnb[i % 56]; S.dfs[i / 56][i % 56] = cd_->ct[i / 56].dfs[i % 56]; }
    if (t < 3) S.log[t] = cd_->ct[t].log;
    for (u32 i = t; i < 256; i += SMALL_G * SMALL_OBJECTS) { S.hlen[i] = cd_->hlen[i]; S.hcode[i] = cd_->hcode[i]; }
    for (u32 i = t; i < SMALL_OBJECTS * SMALL_STAGE_WORDS; i += SMALL_G * SMALL_OBJECTS) (&S.stage[0][0])[i] = 0;
    __syncthreads();
    const u32 c = blockIdx.x * SMALL_OBJECTS + slot;
    if (c >= nchunks) return;
    // (an object's group: LDS traffic of one wave is served in the order it was issued; what must not happen is the compiler moving it)
    auto group_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    // sums and ORs over the group's sixteen lanes
    auto gsum = [&](u32 v) -> u32 { v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64); return v; };
    const zl_chunk_desc cd = a.chunks[c];
    const u32 n = cd.src_size;
    const u32 nseq = a.meta[c].nseq, nlit = a.meta[c].nlit;
    const u8* const src = a.src + cd.src_off;
    const zl_seq* const seqs = a.seqs + (size_t)c * a.seq_stride;
    const u8* const lit = nseq ? a.lits + (size_t)c * a.lit_stride : src;
    const bool ok = cd_->ok != 0 && cd_->huf_ok != 0 && n >= 1 && n < 16384 && nlit <= SMALL_MAX_LIT && nseq <= SMALL_MAX_SEQ && S.log[0] != 0 && S.log[1] != 0 && S.log[2] != 0;
    // ---- the literals' bits under the dictionary's codes; every sequence's codes against what the dictionary's tables can say
    u32 lbits = 0, bad = 0;
    if (ok) {
        for (u32 i = lane; i < nlit; i += SMALL_G) { const u32 l = S.hlen[lit[i]]; lbits += l; bad |= l == 0 ? 1u : 0u; }
        for (u32 i = lane; i < nseq; i += SMALL_G) {
            const zl_seq q = seqs[i];
            const u32 prev_cum = i ? seqs[i - 1].lit_cum : 0u, prev_off = i ? seqs[i - 1].offset : 0u;
            const u32 ll = q.lit_cum - prev_cum;
            const u32 ob = (ll > 0 && q.offset == prev_off) ? 1u : q.offset + 3;  // (repeat offset 1 only, as zl_entropy_kernel below level 5)
            const u32 cl = ll_code(ll), co = hb32(ob), cm = ml_code(q.match_len);
            if (!((cd_->sym_mask[T_LL] >> cl) & 1) || !((cd_->sym_mask[T_OF] >> co) & 1) || !((cd_->sym_mask[T_ML] >> cm) & 1)) bad = 1;
        }
    }
    lbits = gsum(lbits);
    bad = gsum(bad);
    if (!ok || bad) {  // (the same in all of the group's lanes) not this kernel's: the general one takes it
        if (lane == 0) a.frame_size[c] = 0xFFFFFFFFu;
        return;
    }
    u8* const out = a.out + (size_t)c * a.out_stride;
    const u32 fh = (n < 256 ? 6u : 7u) + dict_id_bytes(a.dict_id);
    if (lane == 0) write_frame_header(out, n, 0, a.dict_id);
    u8* const blk = out + fh + 3;
    // ---- literals section
    const u32 csize = (lbits + 8) >> 3;
    const u32 lhs_r = 1 + (nlit >= 32);
    const bool huf = nlit >= 6 && 3 + csize < lhs_r + nlit;
    // Literals the dictionary's codes do not shorten -- an object unlike the dictionary: i.i.d. symbols, numbers -- may well pay for a table of
    // their own: the general kernel weighs one, this one cannot.  (scripts/fuzz_multi.py, seed 161: 300 bytes over ten letters, 314 bytes as a
    // batch item -- a raw block -- against 158 through the one-object entry.)
    if (!huf && nlit >= SMALL_OWN_TABLE_FROM) {
        if (lane == 0) a.frame_size[c] = 0xFFFFFFFFu;
        return;
    }
    u32 lit_sec;
    if (huf) {
        lit_sec = 3 + csize;
        if (lane == 0) wr_le(blk, 3ull | ((u64)nlit << 4) | ((u64)csize << 14), 3);  // Treeless_Literals_Block, one stream, 10-bit sizes
        // the stream (zstd.c:17890 HUF_compress1X): the LAST literal's code in the lowest bits, the first literal's just below the end mark.  A lane
        // takes four literals a round, the round's sixty-four codes are placed by a scan over the group's lanes
        u32* const stage = S.stage[slot];
        u32 at = 0;  // bits placed so far
        for (u32 e0 = 0; e0 < nlit; e0 += 4 * SMALL_G) {
            u64 bits = 0; u32 nb = 0;
#pragma unroll
            for (u32 j = 0; j < 4; j++) {
                const u32 e = e0 + 4 * lane + j;  // emission index: literal nlit - 1 - e
                if (e < nlit) { const u32 sym = lit[nlit - 1 - e]; bits |= (u64)S.hcode[sym] << nb; nb += S.hlen[sym]; }
            }
            u32 inc = nb;
#pragma unroll
            for (int d = 1; d < SMALL_G; d <<= 1) { const u32 x = __shfl_up(inc, d, SMALL_G); if ((int)lane >= d) inc += x; }
            stage_or(stage, at + inc - nb, bits, nb);
            at += (u32)__shfl((int)inc, SMALL_G - 1, SMALL_G);
        }
        if (lane == 0) stage_or(stage, at, 1, 1);  // the end mark
        group_sync();
        u8* const p = blk + 3;
        const u8* const sb = (const u8*)stage;
        for (u32 i = lane; i < csize; i += SMALL_G) p[i] = sb[i];
    } else {
        lit_sec = lhs_r + nlit;
        if (lane == 0) wr_le(blk, 0u | ((lhs_r == 1 ? 0u : 1u) << 2) | ((u64)nlit << (lhs_r == 1 ? 3 : 4)), lhs_r);
        for (u32 i = lane; i < nlit; i += SMALL_G) blk[lhs_r + i] = lit[i];
    }
    // ---- sequences section
    u8* const sq = blk + lit_sec;
    u32 seq_sec = 1;
    if (nseq == 0) { if (lane == 0) sq[0] = 0; }
    else {
        // every lane of the group runs the same chain (zstd.c:21146 ZSTD_encodeSequences), last sequence first; a sequence's record is read from memory by
        // all of them (one address: one transaction), the one in front of it -- which its literal count and its repeat test need -- a step ahead
        u8* const sbp = sq + 2;
        u64 acc = 0; u32 nb = 0, wr = 0;
        auto put = [&](u32 v, u32 bits) { acc |= (u64)(v & ((bits < 32 ? (1u << bits) : 0u) - 1u)) << nb; nb += bits; };
        auto flush = [&]() { if (nb >= 32) { if (lane == 0) { const u32 w = (u32)acc; __builtin_memcpy(sbp + wr, &w, 4); } wr += 4; acc >>= 32; nb -= 32; } };
        u32 st_ll = 0, st_of = 0, st_ml = 0;
        zl_seq cur = seqs[nseq - 1];
        for (u32 k = nseq; k-- > 0;) {
            zl_seq prev; prev.lit_cum = 0; prev.offset = 0; prev.match_len = 0; prev.pos = 0;
            if (k) prev = seqs[k - 1];
            const u32 ll = cur.lit_cum - prev.lit_cum;
            const u32 ob = (ll > 0 && k && cur.offset == prev.offset) ? 1u : cur.offset + 3;
            const u32 cl = ll_code(ll), co = hb32(ob), cm = ml_code(cur.match_len);
            if (k == nseq - 1) {
                st_ml = fse_init_state(S.state[T_ML], S.dnb[T_ML][cm], S.dfs[T_ML][cm]);
                st_of = fse_init_state(S.state[T_OF], S.dnb[T_OF][co], S.dfs[T_OF][co]);
                st_ll = fse_init_state(S.state[T_LL], S.dnb[T_LL][cl], S.dfs[T_LL][cl]);
            } else {
                const u32 eo = fse_encode(S.state[T_OF], S.dnb[T_OF][co], S.dfs[T_OF][co], st_of);
                put(eo & 0xFFFF, eo >> 16); flush();
                const u32 em = fse_encode(S.state[T_ML], S.dnb[T_ML][cm], S.dfs[T_ML][cm], st_ml);
                put(em & 0xFFFF, em >> 16); flush();
                const u32 el = fse_encode(S.state[T_LL], S.dnb[T_LL][cl], S.dfs[T_LL][cl], st_ll);
                put(el & 0xFFFF, el >> 16); flush();
            }
            put(ll, e_LL_bits[cl]); flush();
            put(cur.match_len - 3, e_ML_bits[cm]); flush();
            put(ob, co); flush();
            cur = prev;
        }
        put(st_ml, S.log[T_ML]); flush();
        put(st_of, S.log[T_OF]); flush();
        put(st_ll, S.log[T_LL]); flush();
        put(1, 1);  // the end mark
        const u32 tail = (nb + 7) >> 3;
        if (lane == 0) { for (u32 bq = 0; bq < tail; bq++) sbp[wr + bq] = (u8)(acc >> (8 * bq)); sq[0] = (u8)nseq; sq[1] = 0xFC; }  // Repeat_Mode three times
        seq_sec = 2 + wr + tail;
    }
    // ---- the block: compressed where that is smaller than the content (zstd.c:26880), else raw
    u32 bsize = lit_sec + seq_sec;
    if (bsize >= n) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (u32 i = lane; i < n; i += SMALL_G) blk[i] = src[i];
        if (lane == 0) wr_le(out + fh, ((u64)n << 3) | 1u, 3);
        bsize = n;
    } else if (lane == 0) wr_le(out + fh, ((u64)bsize << 3) | (2u << 1) | 1u, 3);
    if (lane == 0) a.frame_size[c] = fh + 3 + bsize;
}

// exclusive scan of frame sizes: one workgroup of 1024 threads walks tiles of 4096 sizes, four consecutive ones per thread (coalesced), a running
// total from tile to tile.  (Rounds 1-5 gave every thread of 256 a contiguous range: strided loads, 190 us for the 100 000 sizes of config C4's batch.)
#define ZL_SCAN_THREADS 1024
__global__ __launch_bounds__(ZL_SCAN_THREADS) void zl_scan_sizes_kernel(const u32* frame_size, u64* frame_off, const u64* total_in, u64* total_out, u32 n) {
    __shared__ u32 wsum[ZL_SCAN_THREADS / 64];
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    u64 carry = *total_in;  // frames of earlier launch groups come first
    for (u32 base = 0; base < n; base += 4 * ZL_SCAN_THREADS) {
        const u32 i = base + 4 * t;
        u32 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i + k < n ? frame_size[i + k] : 0u;
        const u32 mine = v[0] + v[1] + v[2] + v[3];  // (a tile's total stays below 2^32: 4096 frames of at most a chunk and a header each)
        u32 inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 x = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += x; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 pre = 0, tile = 0;
#pragma unroll
        for (u32 w = 0; w < ZL_SCAN_THREADS / 64; w++) { const u32 x = wsum[w]; if (w < wave) pre += x; tile += x; }
        u64 run = carry + pre + inc - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (i + k < n) frame_off[i + k] = run; run += v[k]; }
        carry += tile;
        __syncthreads();
    }
    if (t == 0) *total_out = carry;
}

__global__ __launch_bounds__(ENC_THREADS) void zl_gather_kernel(zl_gather_args a) {
    const u32 c = blockIdx.x;
    const u8* s = a.staged + (size_t)c * a.out_stride;
    u8* d = a.dst + a.frame_off[c];
    const u32 n = a.frame_size[c];
    copy_wg(d, s, n);
}


// ---------------------------------------------------------------------------------------
// Launch order: longest first.  A chunk of text costs the match kernel eight times what a chunk of random bytes costs (9.3 against
// 1.1 ms per GiB); a launch whose expensive chunks come LAST ends with a few workgroups running while the device drains -- the
// bench mix (a quarter each of random bytes, i.i.d. bytes, text, 20-bit integers, in that order) lost 0.5 of its 3.9 ms that way.
// zl_chunk_cost_kernel looks at 1 KiB from the middle of every chunk: how many of its 1024 positions repeat eight bytes seen
// earlier in the sample (a 4096-entry tag table in LDS, one wave per chunk, sixteen look-up-and-insert rounds) -- match-rich data
// repeats there, match-poor data does not; zl_chunk_order_kernel sorts the chunks by that class, descending, input order inside
// a class (a counting sort on one workgroup).  Only the schedule depends on it: every chunk's bytes are what they were.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void zl_chunk_cost_kernel(const zl_chunk_desc* chunks, const u8* src_base, u32 nchunks, u32* cost) {
    __shared__ u32 tab[4096];
    const u32 c = blockIdx.x, lane = threadIdx.x;
    const zl_chunk_desc cd = chunks[c];
    u32 hits = 0;
    if (cd.src_size >= 4096) {
        for (u32 i = lane; i < 4096; i += 64) tab[i] = 0;
        const u8* p = src_base + cd.src_off + (cd.src_size >> 1) + 16 * lane;  // this lane's sixteen positions: bytes [0, 24)
        const uint4 a = ld128(p);
        const u64 b = ld64(p + 16);
        const u64 w[3] = {(u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), b};
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < 16; r++) {
            const u64 v = r & 7 ? (w[r >> 3] >> (8 * (r & 7))) | (w[(r >> 3) + 1] << (64 - 8 * (r & 7))) : w[r >> 3];
            const u64 h = v * 0x9E3779B185EBCA87ULL;
            const u32 slot = (u32)(h >> 52), tag = (u32)(h >> 20) | 1u;
            hits += tab[slot] == tag ? 1u : 0u;
            __syncthreads();  // (one wave: orders the round's look-ups in front of its inserts)
            tab[slot] = tag;
            __syncthreads();
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) hits += __shfl_xor(hits, d, 64);
    }
    if (lane == 0) cost[c] = hits >= 256 ? 3u : (hits >= 64 ? 2u : (hits >= 8 ? 1u : 0u));
}

#define ZL_ORDER_THREADS 1024
__global__ __launch_bounds__(ZL_ORDER_THREADS) void zl_chunk_order_kernel(const u32* cost, u32 n, u32* order) {
    __shared__ u32 cnt[4][ZL_ORDER_THREADS];  // per class and thread: chunks of that class in the thread's range, then where they go
    const u32 t = threadIdx.x;
    const u32 per = (n + ZL_ORDER_THREADS - 1) / ZL_ORDER_THREADS;
    const u32 b = t * per < n ? t * per : n, e = b + per < n ? b + per : n;
    u32 k[4] = {0, 0, 0, 0};
    for (u32 i = b; i < e; i++) k[cost[i] & 3]++;
    for (int q = 0; q < 4; q++) cnt[q][t] = k[q];
    __syncthreads();
    if (t < 4) {  // class 3 first; inside a class the threads' ranges in order
        u32 base = 0;
        for (u32 q = 3; q > t; q--) for (u32 i = 0; i < ZL_ORDER_THREADS; i++) base += cnt[q][i];  // (a few thousand additions, once per launch)
        for (u32 i = 0; i < ZL_ORDER_THREADS; i++) { const u32 v = cnt[t][i]; cnt[t][i] = base; base += v; }
    }
    __syncthreads();
    u32 at[4] = {cnt[0][t], cnt[1][t], cnt[2][t], cnt[3][t]};
    for (u32 i = b; i < e; i++) order[at[cost[i] & 3]++] = i;
}

extern "C" hipError_t zl_launch_chunk_order(const zl_chunk_desc* chunks, const u8* src, u32 nchunks, u32* cost_scratch, u32* order, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(zl_chunk_cost_kernel, dim3(nchunks), dim3(64), 0, stream, chunks, src, nchunks, cost_scratch);
    hipLaunchKernelGGL(zl_chunk_order_kernel, dim3(1), dim3(ZL_ORDER_THREADS), 0, stream, (const u32*)cost_scratch, nchunks, order);
    return hipGetLastError();
}

extern "C" hipError_t zl_launch_entropy(const zl_entropy_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    if (a->small_done) hipLaunchKernelGGL(zl_entropy_small_kernel, dim3((nchunks + SMALL_OBJECTS - 1) / SMALL_OBJECTS), dim3(SMALL_G * SMALL_OBJECTS), 0, stream, *a, nchunks);
    if (a->handover) {
        hipLaunchKernelGGL(zl_entropy_kernel<1>, dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
        hipLaunchKernelGGL(zl_entropy_kernel<2>, dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    } else
    hipLaunchKernelGGL(zl_entropy_kernel<0>, dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a, nchunks);
    return hipGetLastError();
}
extern "C" hipError_t zl_launch_scan_sizes(const u32* frame_size, u64* frame_off, const u64* total_in, u64* total_out, u32 nchunks, hipStream_t stream) {
    hipLaunchKernelGGL(zl_scan_sizes_kernel, dim3(1), dim3(ZL_SCAN_THREADS), 0, stream, frame_size, frame_off, total_in, total_out, nchunks);
    return hipGetLastError();
}
extern "C" hipError_t zl_launch_gather(const zl_gather_args* a, u32 nchunks, hipStream_t stream) {
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(zl_gather_kernel, dim3(nchunks), dim3(ENC_THREADS), 0, stream, *a);
    return hipGetLastError();
}

// debug: what the runtime says about resident workgroups per CU (not part of include/zl_hip.h)
extern "C" int zl_debug_occupancy(int which) {
    int n = -1;
    if (which == 0) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zl_match_auto_kernel<false, false, false, false>, ENC_THREADS, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, zl_entropy_kernel<0>, ENC_THREADS, 0);
    return n;
}

extern "C" hipError_t zl_launch_dict_hash(const u8* d_content, u32 n, u32* d_tables, hipStream_t stream) {
    hipLaunchKernelGGL(zl_dict_hash_kernel, dim3(1), dim3(ENC_THREADS), 0, stream, d_content, n, d_tables);
    return hipGetLastError();
}

extern "C" hipError_t zl_launch_cdict_build(const void* d_dict_tables, void* d_cdict, hipStream_t stream) {
    hipLaunchKernelGGL(zl_cdict_build_kernel, dim3(1), dim3(ENC_THREADS), 0, stream, (const zl_dict_tables*)d_dict_tables, (zl_cdict*)d_cdict);
    return hipGetLastError();
}
