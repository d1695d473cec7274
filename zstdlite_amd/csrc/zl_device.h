// zl_device.h -- small device-side helpers shared by the HIP kernels: little-endian reads and
// the two bit-readers of the zstd format (forward LSB-first for table descriptions, backward
// for entropy streams; zstd.c:2390-2470 BIT_initDStream/BIT_lookBits describe the latter).
#pragma once
#include <hip/hip_runtime.h>

#include "zl_format.h"

__device__ __forceinline__ u32 zl_rd_le(const u8* p, u32 nbytes) {
    u32 v = 0;
    for (u32 i = 0; i < nbytes; i++) v |= (u32)p[i] << (8 * i);
    return v;
}

// 64 stream bits starting at byte index `byte` (< len); bytes past the end read as zero and
// nothing outside [p, p+len) is touched.
__device__ __forceinline__ u64 zl_window64(const u8* p, u32 len, u32 byte) {
    u64 w;
    if (byte + 8 <= len) {
        __builtin_memcpy(&w, p + byte, 8);
        return w;
    }
    if (len >= 8) {
        __builtin_memcpy(&w, p + len - 8, 8);
        return w >> (8 * (byte - (len - 8)));
    }
    w = 0;
    for (u32 i = byte; i < len; i++) w |= (u64)p[i] << (8 * (i - byte));
    return w;
}

// n (<= 32) bits whose lowest bit is stream bit `pos`; bit i of the stream is bit (i & 7) of
// byte i >> 3.  Bits at negative positions (a state update that runs past the start of a
// backward stream) and bits past the end read as zero.
__device__ __forceinline__ u32 zl_bits_bwd(const u8* p, u32 len, i32 pos, u32 n) {
    if (n == 0) return 0;
    const u64 mask = (1ull << n) - 1;
    if (pos >= 0) {
        const u32 byte = (u32)pos >> 3;
        if (byte >= len) return 0;
        return (u32)((zl_window64(p, len, byte) >> (pos & 7)) & mask);
    }
    const i32 avail = (i32)n + pos;
    if (avail <= 0) return 0;
    return (u32)((zl_window64(p, len, 0) & ((1ull << avail) - 1)) << (u32)(-pos));
}

__device__ __forceinline__ u32 zl_bits_fwd(const u8* p, u32 len, i32 pos, u32 n) { return zl_bits_bwd(p, len, pos, n); }

// Backward stream: the last byte carries a 1-bit end marker above the payload.  Returns the
// number of payload bits, or -1 when the stream is empty / the marker is missing.
__device__ __forceinline__ i32 zl_bwd_init(const u8* p, u32 len) {
    if (len == 0) return -1;
    const u32 last = p[len - 1];
    if (last == 0) return -1;
    return (i32)(len - 1) * 8 + (31 - __clz(last));
}

// ---------------------------------------------------------------------------------------
// Register-buffered backward bit reader for the serial entropy chains.  The stream is viewed as
// aligned 64-bit words; a 128-bit window (hi:lo) plus two prefetched words travel down the
// stream, so the global-memory latency of a refill is paid two words (128+ stream bits) before
// the data is needed instead of on every field.  Reads must be issued at non-increasing bit
// positions (which is how a backward zstd stream is consumed).  Positions below zero and the
// rare case of a field straddling the start of the stream fall back to zl_bits_bwd.
// The word holding the last stream byte may extend up to 7 bytes past it: source buffers must
// be readable up to the next 8-byte boundary (device allocations always are).
// ---------------------------------------------------------------------------------------
struct zl_bwd_reader {
    const u64* w;   // aligned word 0 contains stream byte 0
    const u8* p;    // stream start (for the slow path)
    u32 len;
    u32 bitofs;     // bit index of stream bit 0 inside word 0
    i32 k;          // lo = w[k], hi = w[k+1]
    u64 hi, lo, p1, p2;
};

__device__ __forceinline__ void zl_bwd_open(zl_bwd_reader& r, const u8* p, u32 len, i32 top_bits) {
    const uintptr_t a = (uintptr_t)p;
    r.w = (const u64*)(a & ~(uintptr_t)7);
    r.p = p;
    r.len = len;
    r.bitofs = (u32)(a & 7) * 8;
    const i32 top = (i32)r.bitofs + (top_bits > 0 ? top_bits - 1 : 0);  // highest bit that will be read
    const i32 kt = top >> 6;
    r.k = kt > 0 ? kt - 1 : 0;
    r.lo = r.w[r.k];
    r.hi = kt > 0 ? r.w[r.k + 1] : 0;
    r.p1 = r.k >= 1 ? r.w[r.k - 1] : 0;
    r.p2 = r.k >= 2 ? r.w[r.k - 2] : 0;
}

// n <= 32 bits whose lowest bit is stream bit `pos`
__device__ __forceinline__ u32 zl_bwd_read(zl_bwd_reader& r, i32 pos, u32 n) {
    if (n == 0) return 0;
    if (pos < 0) return zl_bits_bwd(r.p, r.len, pos, n);
    const i32 a = (i32)r.bitofs + pos;
    while ((a >> 6) < r.k) {
        r.hi = r.lo;
        r.lo = r.p1;
        r.p1 = r.p2;
        r.k--;
        r.p2 = r.k >= 2 ? r.w[r.k - 2] : 0;
    }
    const u32 sh = (u32)(a - (r.k << 6));
    u64 v;
    if (sh < 64) v = (r.lo >> sh) | (sh ? (r.hi << (64 - sh)) : 0);
    else v = r.hi >> (sh - 64);
    return (u32)v & (u32)((1ull << n) - 1);
}

// ---------------------------------------------------------------------------------------
// Left-aligned accumulator over a backward stream, refilled 32 bits at a time from aligned
// dwords: the next unread stream bit is always the top bit of `acc`, so a peek is one shift of
// the high dword and a consume is one 64-bit shift -- no per-field address arithmetic and no
// branches in the hot loops that use it.  `pos` keeps the exact count of unread payload bits
// so callers can finish a stream with the generic reader and check that it ends on bit 0.
// Bits "below" the stream start (garbage from the enclosing dword, or zeros once idx < 0) can
// be peeked but a valid stream never consumes them; consuming them drives pos negative, which
// every caller treats as corruption.
// ---------------------------------------------------------------------------------------
struct zl_lreader {
    const u32* w;  // aligned dword 0 contains stream byte 0
    u64 acc;
    i32 nav;       // valid bits in acc
    i32 idx;       // next dword to refill from (refills below 0 supply zeros)
    i32 pos;       // unread payload bits
};

// requires top_bits >= 64
__device__ __forceinline__ void zl_lr_open(zl_lreader& r, const u8* p, u32 len, i32 top_bits) {
    const u32 mis = (u32)((uintptr_t)p & 3);
    r.w = (const u32*)(p - mis);  // pointer arithmetic (not an integer round trip) keeps the global address space
    const u32 a_top = mis * 8 + (u32)top_bits;
    const u32 nav0 = 32 + (a_top & 31);
    const u64 top64 = ((u64)zl_bits_bwd(p, len, top_bits - 32, 32) << 32) | zl_bits_bwd(p, len, top_bits - 64, 32);
    r.acc = top64 & ~((1ull << (64 - nav0)) - 1);
    r.nav = (i32)nav0;
    r.idx = (i32)(a_top >> 5) - 2;
    r.pos = top_bits;
}
__device__ __forceinline__ u32 zl_lr_peek(const zl_lreader& r, u32 n) {  // 1 <= n <= 32, needs nav >= n
    return (u32)(r.acc >> 32) >> (32 - n);
}
__device__ __forceinline__ void zl_lr_skip(zl_lreader& r, u32 n) {
    r.acc <<= n;
    r.nav -= (i32)n;
    r.pos -= (i32)n;
}
__device__ __forceinline__ u32 zl_lr_take(zl_lreader& r, u32 n) {  // 0 <= n <= 32
    const u32 v = n ? zl_lr_peek(r, n) : 0;
    zl_lr_skip(r, n);
    return v;
}
__device__ __forceinline__ void zl_lr_refill(zl_lreader& r) {  // call when nav <= 32
    const u32 wv = r.idx >= 0 ? r.w[r.idx] : 0;
    r.acc |= (u64)wv << (32 - r.nav);
    r.nav += 32;
    r.idx--;
}

// up to 64 bits whose lowest bit is stream bit `pos` (pos >= 0)
__device__ __forceinline__ u64 zl_bwd_read64(zl_bwd_reader& r, i32 pos, u32 n) {
    if (n == 0) return 0;
    const i32 a = (i32)r.bitofs + pos;
    while ((a >> 6) < r.k) {
        r.hi = r.lo;
        r.lo = r.p1;
        r.p1 = r.p2;
        r.k--;
        r.p2 = r.k >= 2 ? r.w[r.k - 2] : 0;
    }
    const u32 sh = (u32)(a - (r.k << 6));
    u64 v;
    if (sh < 64) v = (r.lo >> sh) | (sh ? (r.hi << (64 - sh)) : 0);
    else v = r.hi >> (sh - 64);
    return n >= 64 ? v : (v & ((1ull << n) - 1));
}

// ---------------------------------------------------------------------------------------
// XXH64 (zstd.c:11524-11800); any workgroup size >= 4.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 zl_xxh_round(u64 acc, u64 in) {
    acc += in * 0xC2B2AE3D27D4EB4FULL;
    acc = (acc << 31) | (acc >> 33);
    return acc * 0x9E3779B185EBCA87ULL;
}

// XXH64 of the regenerated frame (zstd.c:11708-11800), low 32 bits go in the frame footer.
// The 4 accumulator lanes are inherently serial; the workgroup stages 8 KiB tiles into LDS
// with coalesced loads and lanes 0..3 of wave 0 run the rounds out of LDS.
__device__ inline u32 zl_xxh64_low32(u8* tile /* 8 KiB LDS */, u64* acc4 /* LDS u64[4] */, u32* bcast /* LDS u32 */, const u8* data, u64 len) {
    const u64 P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
              P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    const u32 t = threadIdx.x;
    if (t == 0) { acc4[0] = P1 + P2; acc4[1] = P2; acc4[2] = 0; acc4[3] = 0 - P1; }
    __syncthreads();
    const u64 stripes = len / 32;
    for (u64 s0 = 0; s0 < stripes; s0 += 256) {
        const u32 ns = (u32)((stripes - s0 < 256) ? (stripes - s0) : 256);
        const u8* base = data + s0 * 32;
        for (u32 i = t; i < ns * 32; i += blockDim.x) tile[i] = base[i];
        __syncthreads();
        if (t < 4) {
            u64 acc = acc4[t];
            for (u32 r = 0; r < ns; r++) {
                u64 v;
                __builtin_memcpy(&v, tile + r * 32 + t * 8, 8);
                acc = zl_xxh_round(acc, v);
            }
            acc4[t] = acc;
        }
        __syncthreads();
    }
    u32 out = 0;
    if (t == 0) {
        u64 h;
        if (len >= 32) {
            u64 v1 = acc4[0], v2 = acc4[1], v3 = acc4[2], v4 = acc4[3];
            h = ((v1 << 1) | (v1 >> 63)) + ((v2 << 7) | (v2 >> 57)) + ((v3 << 12) | (v3 >> 52)) + ((v4 << 18) | (v4 >> 46));
            h = (h ^ zl_xxh_round(0, v1)) * P1 + P4;
            h = (h ^ zl_xxh_round(0, v2)) * P1 + P4;
            h = (h ^ zl_xxh_round(0, v3)) * P1 + P4;
            h = (h ^ zl_xxh_round(0, v4)) * P1 + P4;
        } else {
            h = P5;
        }
        h += len;
        const u8* p = data + stripes * 32;
        const u8* end = data + len;
        while (end - p >= 8) {
            u64 v = 0;
            for (int i = 0; i < 8; i++) v |= (u64)p[i] << (8 * i);
            h ^= zl_xxh_round(0, v);
            h = ((h << 27) | (h >> 37)) * P1 + P4;
            p += 8;
        }
        if (end - p >= 4) {
            u64 v = 0;
            for (int i = 0; i < 4; i++) v |= (u64)p[i] << (8 * i);
            h ^= v * P1;
            h = ((h << 23) | (h >> 41)) * P2 + P3;
            p += 4;
        }
        while (p < end) {
            h ^= (u64)(*p++) * P5;
            h = ((h << 11) | (h >> 53)) * P1;
        }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        out = (u32)h;
        bcast[0] = out;
    }
    __syncthreads();
    out = bcast[0];
    __syncthreads();
    return out;
}

// Workgroup w works on item zl_spread(w, n): pseudo-random order inside windows of 256 items, input
// order across windows.  Measured on the 1 GiB mix (match kernel): input order is best when similar
// items sit together (9.1 ms; a fully pseudo-random order lets expensive stragglers start late:
// 10.6 ms), but costs 50 % when the cost alternates with a small period (four kinds of data
// alternating chunk by chunk: 15.2 ms; windows of 64 do not cure it, windows of 256 give 10.4).
__device__ __forceinline__ u32 zl_spread(u32 w, u32 n) {
#ifndef ZL_SPREAD_LOG
#define ZL_SPREAD_LOG 8
#endif
    const u32 mask = (1u << ZL_SPREAD_LOG) - 1;
    if ((w | mask) >= n) return w;  // the ragged last window stays as it is
    u32 x = w & mask;               // pseudo-random order inside a window, input order across windows
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> (ZL_SPREAD_LOG / 2);
    x = (x * 0x85EBCA77u) & mask;
    x ^= x >> (ZL_SPREAD_LOG / 2);
    return (w & ~mask) | x;
}

// Debug aid (-DZL_POISON_LDS=<mask of kernels: 1 entropy decode, 2 exec, 4 match, 8 entropy encode>): kernels fill their LDS with a pattern before use, which turns a read of
// never-written LDS -- normally whatever the previous workgroup left there -- into a reproducible failure.
#ifdef ZL_POISON_LDS
#define ZL_POISON(obj_, bit_) do { if (!((ZL_POISON_LDS) & (bit_))) break; u32* p__ = (u32*)&(obj_); for (u32 i__ = threadIdx.x; i__ < sizeof(obj_) / 4; i__ += blockDim.x) p__[i__] = 0xEEEEEEEEu; __syncthreads(); } while (0)
#else
#define ZL_POISON(obj_, bit_) do { } while (0)
#endif

// Workgroup barrier for data handed from one wave to another THROUGH GLOBAL MEMORY inside a kernel.
// On gfx950 __syncthreads() waits for LDS traffic only (the compiler's memory model counts on the
// CU's L1 seeing the workgroup's own stores in order); a wave that stored and then passed the barrier
// may still have those stores in flight.  Draining them first costs one wait per hand-off and removed
// a rare (about 1 in 15 fresh-box runs) wrong byte in the 1 GiB round trip.
__device__ __forceinline__ void zl_sync_global() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
