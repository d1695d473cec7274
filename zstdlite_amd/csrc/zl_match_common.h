// zl_match_common.h -- constants and small helpers of the match kernel (zl_encode.hip): hashes, byte compares, the
// LDS-only barrier.
#pragma once
#include "zl_device.h"

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
#define SHORT_LOG_BULK 11  // the bulk instantiation of the match kernel (zl_encode.hip says why)
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
