// zl_match_common.h -- constants and small helpers of the match kernel (zl_encode.hip): hashes, byte compares, the
// LDS-only barrier.
#pragma once
#include "zl_device.h"
#include "zl_encode.h"

#ifndef LONG_LOG
#define LONG_LOG 13        // 8-byte-hash table: 2^13 16-bit entries (16 KiB LDS)
#endif
#ifndef SHORT_LOG
#define SHORT_LOG 12       // 6-byte-hash table: 2^12 16-bit entries (8 KiB LDS)
#endif
#ifndef LANE_MATCH_CAP
#define LANE_MATCH_CAP 40
#endif
#ifndef LANE_MATCH_CAP_FAR
#define LANE_MATCH_CAP_FAR 72  // the windowed layout verifies further: python sources +0.9 % ratio for 5 % of match time
#endif
// LANE_MATCH_CAP:  // per-lane verification stops here; longer matches continue as repeat-offset pieces and are merged in E2
#define MIN_MATCH 4

__device__ __forceinline__ u64 ld64(const u8* p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint4 ld128(const u8* p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
#ifndef ZL_LONG_BYTES
#define ZL_LONG_BYTES 8
#endif
#ifndef ZL_SHORT_BYTES
#define ZL_SHORT_BYTES 6
#endif
__device__ __forceinline__ u32 hash8(u64 v) { return (u32)(((v << (64 - 8 * ZL_LONG_BYTES)) * 0x9E3779B185EBCA87ULL) >> (64 - LONG_LOG)); }
__device__ __forceinline__ u32 hash5(u64 v) { return (u32)(((v << (64 - 8 * ZL_SHORT_BYTES)) * 0xC2B2AE3D27D4EB4FULL) >> (64 - SHORT_LOG)); }
// the windowed layout hashes 5 bytes, the reference's minimum match at level 3 (zstd.c:29529): on real text the
// 5-byte repeats are worth more than the table pressure they cost (python sources: +2.4 % ratio); the synthetic
// bench inputs prefer 6 (fewer, longer candidates: faster and no worse)
#ifndef ZL_SHORT_BYTES_FAR
#define ZL_SHORT_BYTES_FAR 5
#endif
#ifndef LONG_LOG_HI
#define LONG_LOG_HI 14     // levels >= 5
#endif
#ifndef SHORT_LOG_HI
#define SHORT_LOG_HI 13
#endif
__device__ __forceinline__ u32 hash_long(u64 v, u32 shift /* 64 - log2 of the table's entries */) { return (u32)(((v << (64 - 8 * ZL_LONG_BYTES)) * 0x9E3779B185EBCA87ULL) >> shift); }
#ifndef SHORT_LOG_FAR
#define SHORT_LOG_FAR 12   // the one-frame layout at level 3: 2^13 + 2^12 entries = 27 KiB of LDS with the rest, six workgroups per CU (round 3: 2^13 + 2^13, four per CU).
                           // Round 4, 1 GiB mix in one frame: match 7.55 -> 7.07 ms for 0.9 % of the python sources (4.390 -> 4.353; the reference's level 3: 4.428)
#endif
#ifndef SHORT_LOG_BULK
#define SHORT_LOG_BULK 12  // the bulk instantiation of the match kernel (zl_encode.hip says why; 11 until round 6, beside a long table of 2^13)
#endif
template <bool FAR>
__device__ __forceinline__ u32 hash_short(u64 v, u32 shift /* 64 - log2 of the table's entries */) {
    return (u32)(((v << (64 - 8 * (FAR ? ZL_SHORT_BYTES_FAR : ZL_SHORT_BYTES))) * 0xC2B2AE3D27D4EB4FULL) >> shift);
}
// (the byte count as a shift that the caller holds in a register: 64 - 8 * bytes)
__device__ __forceinline__ u32 hash_short_rt(u64 v, u32 byte_shift, u32 shift) { return (u32)(((v << byte_shift) * 0xC2B2AE3D27D4EB4FULL) >> shift); }
__device__ __forceinline__ u32 common_bytes(u64 a, u64 b) {
    const u64 x = a ^ b;
    return x ? (u32)(__builtin_ctzll(x) >> 3) : 8u;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// wait for every literal / sequence store of the emit phase at each of the ~14 barriers of an
// iteration; nothing in the window loop hands global data from one thread to another.  RULE: any code of
// this kernel that reads global memory another thread of the workgroup wrote (today: only the merge pass
// behind the loop, which re-reads the sequences) must use zl_sync_global() -- vmcnt(0) + barrier -- not
// this; tests/test_fullsize_gpu.py::test_one_gib_ten_times_on_fresh_contexts is the net under it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Which nodes of a wave's 64 lie on the path that enters at `entry` (a lane of this wave, or none)?  jr[r] = the node 2^r steps
// on (window-relative; >= wbase + 64: outside).  Every node that is marked marks the node 2^r steps on, r = 0..5: the marks
// travel from lane to lane through ds_permute (each lane SENDS a 1 to the lane it names; lanes that have nothing to say name
// lane 0, which no step can land on, and whatever arrives there is ignored) -- six LDS-crossbar round trips and no memory.
// Rounds 1-3 kept the marks in an LDS byte array: a read and a dependent write per round, fourteen round trips.
__device__ __forceinline__ bool zl_path_marks(const u32 (&jr)[6], u32 t, u32 wbase, u32 my_entry) {
    const u32 lane = t & 63;
    u32 mk = (t == my_entry) ? 1u : 0u;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const bool send = mk != 0 && jr[r] < wbase + 64;
        const u32 dst = send ? ((jr[r] & 63) << 2) : 0u;
        const u32 got = (u32)__builtin_amdgcn_ds_permute((int)dst, 1);
        mk |= lane ? got : 0u;
    }
    return mk != 0;
}

// ---- merge continuation pieces.  A lane verifies a bounded number of bytes, so a longer match arrives as several sequences:
// one starts where the last one ends, with the same offset and no literal between them.  Coded as they are, every piece after
// the first would spell its offset out (a repeat offset needs a literal in front of it, zstd.c:19680); joined, they are the one
// sequence the reference would have found.  One pass over the chunk's sequences, 256 at a time, compacting in place: heads move
// down, pieces add their length to their head.  t_end, t_off: 256 words of LDS each; scan_a: 4 words.  Returns the new count.
__device__ __forceinline__ u32 zl_merge_pieces(zl_seq* seqs, u32 nseq, u32* t_end, u32* t_off, u32* scan_a) {
    const u32 t = threadIdx.x;
    u32 out_n = nseq;
    if (nseq > 1) {
        zl_sync_global();  // the sequences were written by other lanes
        u32 carried_end = 0xFFFFFFFFu, carried_off = 0;  // of the last sequence of the tile before (uniform)
        out_n = 0;
        for (u32 base = 0; base < nseq; base += 256) {
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
            const u32 wave = t >> 6, wh = (u32)__popcll(mh);
            if ((t & 63) == 0) scan_a[wave] = wh;
            const u32 last = (nseq - base < 256 ? nseq - base : 256) - 1;
            const u32 next_end = t_end[last], next_off = t_off[last];
            lds_barrier();
            u32 pre = 0, tot = 0;
#pragma unroll
            for (u32 w = 0; w < 4; w++) { const u32 x = scan_a[w]; if (w < wave) pre += x; tot += x; }
            // heads up to and including this lane: the head's new slot is that count minus one
            const u32 dest = out_n + pre + (u32)__popcll(mh & ((2ull << (t & 63)) - 1)) - 1;
            if (head) seqs[dest] = sq;
            zl_sync_global();  // heads are in place before their pieces add to them
            if (cont) atomicAdd(&seqs[dest].match_len, sq.match_len);
            carried_end = next_end;
            carried_off = next_off;
            out_n += tot;
            lds_barrier();
        }
    }
    return out_n;
}
