// zl_xxh64.h -- XXH64 on the HOST, streaming (zstd.c:11524-11800 XXH64_reset / XXH64_update / XXH64_digest; the frame
// footer carries its low 32 bits, zstd.c:27733 ZSTD_writeEpilogue, checked at zstd.c:41646-41660).
//
// Why the host: the content checksum of ONE frame is one chain through the whole content -- four accumulators that each
// depend on their previous value every 32 bytes, a rotate between the add and the multiply, so neither a prefix scan nor
// a split over workgroups exists.  A CPU core runs the chain at 15-20 GB/s; four GPU lanes at well under 1 GB/s (a
// 64-bit multiply is four quarter-rate instructions there).  Frames of at most one chunk are still checked on the device,
// thousands of them side by side (zl_device.h zl_xxh64_low32).  The host entries already hold the content in host memory:
// a thread hashes it beside the upload (compression) or beside the download (decompression).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace zlx {

static const uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;

static inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }  // (little-endian hosts only: x86-64 here)
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t round1(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
static inline uint64_t merge(uint64_t h, uint64_t v) { return (h ^ round1(0, v)) * P1 + P4; }

struct Xxh64 {
    uint64_t v[4];
    uint64_t total;
    uint8_t buf[32];
    uint32_t fill;
    Xxh64() { reset(); }
    void reset() { v[0] = P1 + P2; v[1] = P2; v[2] = 0; v[3] = 0 - P1; total = 0; fill = 0; }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        total += n;
        if (fill) {
            const size_t k = n < 32 - fill ? n : 32 - fill;
            memcpy(buf + fill, p, k);
            fill += (uint32_t)k; p += k; n -= k;
            if (fill < 32) return;
            for (int i = 0; i < 4; i++) v[i] = round1(v[i], rd64(buf + 8 * i));
            fill = 0;
        }
        uint64_t a = v[0], b = v[1], c = v[2], d = v[3];
        while (n >= 32) {
            a = round1(a, rd64(p)); b = round1(b, rd64(p + 8)); c = round1(c, rd64(p + 16)); d = round1(d, rd64(p + 24));
            p += 32; n -= 32;
        }
        v[0] = a; v[1] = b; v[2] = c; v[3] = d;
        if (n) { memcpy(buf, p, n); fill = (uint32_t)n; }
    }
    uint64_t digest() const {
        uint64_t h;
        if (total >= 32) {
            h = rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18);
            h = merge(h, v[0]); h = merge(h, v[1]); h = merge(h, v[2]); h = merge(h, v[3]);
        } else {
            h = P5;  // (seed 0)
        }
        h += total;
        const uint8_t* p = buf;
        uint32_t n = fill;
        while (n >= 8) { h ^= round1(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; n -= 8; }
        if (n >= 4) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; n -= 4; }
        while (n) { h ^= (uint64_t)(*p++) * P5; h = rotl(h, 11) * P1; n--; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        return h;
    }
    uint32_t low32() const { return (uint32_t)digest(); }
};

}  // namespace zlx
