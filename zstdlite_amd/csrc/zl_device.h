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
