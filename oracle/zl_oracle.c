/*
 * zl_oracle.c -- TEST INFRASTRUCTURE ONLY (see zl_oracle.h).
 *
 * A from-scratch, clarity-first CPU restatement of the zstd frame decoder
 * (RFC 8878 semantics as pinned by the reference's bundled libzstd 1.5.6).
 * Every bit is read through one slow, obviously-correct helper; there are no
 * fast paths.  Each function cites the reference function it restates.
 */
#include "zl_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

#define ERR(c) ((size_t)0 - (size_t)(c))
#define IS_ERR(x) ((x) > ERR(ZLO_ERR_MAXCODE))

#define ZSTD_MAGIC 0xFD2FB528u
#define ZSTD_MAGIC_SKIPPABLE 0x184D2A50u
#define ZSTD_MAGIC_DICT 0xEC30A437u
#define BLOCK_MAX (128u << 10)
#define WINDOWLOG_MAX 31

unsigned zlo_is_error(size_t code) { return IS_ERR(code); }

/* zstd.c:3585-3628 ERR_getErrorString: user-visible strings are part of the contract
 * (tests/testthat/test-checksums.R:24-27 matches "doesn't match checksum"). */
const char *zlo_error_name(size_t code) {
    if (!IS_ERR(code)) return "No error detected";
    switch ((int)(0 - code)) {
    case ZLO_ERR_GENERIC: return "Error (generic)";
    case ZLO_ERR_PREFIX_UNKNOWN: return "Unknown frame descriptor";
    case ZLO_ERR_FRAMEPARAM_UNSUPPORTED: return "Unsupported frame parameter";
    case ZLO_ERR_WINDOW_TOO_LARGE: return "Frame requires too much memory for decoding";
    case ZLO_ERR_CORRUPTION: return "Data corruption detected";
    case ZLO_ERR_CHECKSUM_WRONG: return "Restored data doesn't match checksum";
    case ZLO_ERR_LITERALS_HEADER: return "Header of Literals' block doesn't respect format specification";
    case ZLO_ERR_DICT_CORRUPTED: return "Dictionary is corrupted";
    case ZLO_ERR_DICT_WRONG: return "Dictionary mismatch";
    case ZLO_ERR_TABLELOG_TOO_LARGE: return "tableLog requires too much memory : unsupported";
    case ZLO_ERR_MAXSYMBOL_TOO_LARGE: return "Unsupported max Symbol Value : too large";
    case ZLO_ERR_MAXSYMBOL_TOO_SMALL: return "Specified maxSymbolValue is too small";
    case ZLO_ERR_DSTSIZE_TOO_SMALL: return "Destination buffer is too small";
    case ZLO_ERR_SRCSIZE_WRONG: return "Src size is incorrect";
    default: return "Unspecified error code";
    }
}

/* zstd.h:231 ZSTD_COMPRESSBOUND */
size_t zlo_compress_bound(size_t n) {
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

/* ------------------------------------------------------------------ little helpers */
static u32 rd_le(const u8 *p, int nbytes) {
    u32 v = 0;
    for (int i = 0; i < nbytes; i++) v |= (u32)p[i] << (8 * i);
    return v;
}
static u64 rd_le64(const u8 *p) { return (u64)rd_le(p, 4) | ((u64)rd_le(p + 4, 4) << 32); }
static int highbit(u32 v) { /* floor(log2(v)), v > 0 */
    int r = 0;
    while (v >>= 1) r++;
    return r;
}

/* Read `n` bits (n <= 32) whose lowest bit is stream bit `pos` (bit i of the stream is
 * bit (i&7) of byte i>>3).  Bits below 0 read as zero -- this is what libzstd's
 * backward reader effectively does when a state update runs past the start of the stream
 * (BIT_lookBits on a zero-extended container, zstd.c:2420-2470). */
static u32 bits_at(const u8 *p, i64 pos, int n) {
    u32 v = 0;
    for (int i = 0; i < n; i++) {
        i64 b = pos + i;
        if (b >= 0) v |= (u32)((p[b >> 3] >> (b & 7)) & 1) << i;
    }
    return v;
}

/* Backward bitstream (zstd.c:2390-2410 BIT_initDStream): the last byte holds a 1-bit
 * end marker above the payload.  Returns number of payload bits or -1 if malformed. */
static i64 bwd_init(const u8 *p, size_t n) {
    if (n == 0 || p[n - 1] == 0) return -1;
    return (i64)(n - 1) * 8 + highbit(p[n - 1]);
}

/* ------------------------------------------------------------------ XXH64 */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL
static u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static u64 xxh_round(u64 acc, u64 in) { return rotl64(acc + in * P2, 31) * P1; }
static u64 xxh_merge(u64 h, u64 v) { return (h ^ xxh_round(0, v)) * P1 + P4; }
/* zstd.c:11524-11800 (XXH64_round / XXH64_digest): the algorithm is the public xxHash64
 * definition. */
uint64_t zlo_xxh64(const void *data, size_t len, uint64_t seed) {
    const u8 *p = (const u8 *)data, *end = p + len;
    u64 h;
    if (len >= 32) {
        u64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        while (end - p >= 32) {
            v1 = xxh_round(v1, rd_le64(p));
            v2 = xxh_round(v2, rd_le64(p + 8));
            v3 = xxh_round(v3, rd_le64(p + 16));
            v4 = xxh_round(v4, rd_le64(p + 24));
            p += 32;
        }
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    } else {
        h = seed + P5;
    }
    h += (u64)len;
    while (end - p >= 8) { h ^= xxh_round(0, rd_le64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (end - p >= 4) { h ^= (u64)rd_le(p, 4) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (u64)(*p++) * P5; h = rotl64(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------ frame header */
typedef struct {
    u64 content_size; /* ZLO_CONTENTSIZE_UNKNOWN if absent */
    u64 window_size;
    u32 dict_id;
    int has_checksum;
    int skippable;
    size_t header_size;
    u32 skippable_size;
} frame_hdr;

/* zstd.c:41050-41148 ZSTD_getFrameHeader_advanced.  Returns 0 or an error code. */
static size_t parse_frame_header(frame_hdr *h, const u8 *src, size_t n) {
    memset(h, 0, sizeof(*h));
    if (n < 5) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    u32 magic = rd_le(src, 4);
    if ((magic & 0xFFFFFFF0u) == ZSTD_MAGIC_SKIPPABLE) {
        if (n < 8) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        h->skippable = 1;
        h->skippable_size = rd_le(src + 4, 4);
        h->header_size = 8;
        return 0;
    }
    if (magic != ZSTD_MAGIC) return ERR(ZLO_ERR_PREFIX_UNKNOWN);
    u8 fhd = src[4];
    int dict_flag = fhd & 3, checksum = (fhd >> 2) & 1, single = (fhd >> 5) & 1, fcs_flag = fhd >> 6;
    if (fhd & 0x08) return ERR(ZLO_ERR_FRAMEPARAM_UNSUPPORTED); /* reserved bit */
    static const int dict_bytes[4] = {0, 1, 2, 4};
    static const int fcs_bytes[4] = {0, 2, 4, 8};
    size_t hs = 5 + (single ? 0 : 1) + dict_bytes[dict_flag] + fcs_bytes[fcs_flag] + ((single && !fcs_flag) ? 1 : 0);
    if (n < hs) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    size_t pos = 5;
    u64 window = 0;
    if (!single) {
        u8 wd = src[pos++];
        u32 wlog = (wd >> 3) + 10;
        if (wlog > WINDOWLOG_MAX) return ERR(ZLO_ERR_WINDOW_TOO_LARGE);
        window = 1ULL << wlog;
        window += (window >> 3) * (wd & 7);
    }
    h->dict_id = rd_le(src + pos, dict_bytes[dict_flag]);
    pos += dict_bytes[dict_flag];
    u64 fcs = ZLO_CONTENTSIZE_UNKNOWN;
    switch (fcs_flag) {
    case 0: if (single) fcs = src[pos]; break;
    case 1: fcs = (u64)rd_le(src + pos, 2) + 256; break;
    case 2: fcs = rd_le(src + pos, 4); break;
    default: fcs = rd_le64(src + pos); break;
    }
    if (single) window = fcs;
    h->content_size = fcs;
    h->window_size = window;
    h->has_checksum = checksum;
    h->header_size = hs;
    return 0;
}

unsigned long long zlo_get_frame_content_size(const void *src, size_t n) {
    frame_hdr h;
    if (IS_ERR(parse_frame_header(&h, (const u8 *)src, n))) return ZLO_CONTENTSIZE_ERROR;
    if (h.skippable) return 0; /* zstd.c:41176: skippable frames report 0 */
    return h.content_size;
}

unsigned zlo_frame_dict_id(const void *src, size_t n) {
    frame_hdr h;
    if (IS_ERR(parse_frame_header(&h, (const u8 *)src, n)) || h.skippable) return 0;
    return h.dict_id;
}

/* zstd.c:41335-41398 ZSTD_findFrameSizeInfo: serial walk over 3-byte block headers
 * (zstd.c:43075 ZSTD_getcBlockSize). */
size_t zlo_find_frame_compressed_size(const void *src_, size_t n) {
    const u8 *src = (const u8 *)src_;
    frame_hdr h;
    size_t e = parse_frame_header(&h, src, n);
    if (IS_ERR(e)) return e;
    if (h.skippable) {
        u64 total = 8ULL + h.skippable_size;
        if (total > n) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        return (size_t)total;
    }
    size_t pos = h.header_size;
    for (;;) {
        if (n - pos < 3) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        u32 bh = rd_le(src + pos, 3);
        int last = bh & 1, type = (bh >> 1) & 3;
        u32 bsize = bh >> 3;
        if (type == 3) return ERR(ZLO_ERR_CORRUPTION);
        size_t csize = (type == 1) ? 1 : bsize;
        pos += 3;
        if (n - pos < csize) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        pos += csize;
        if (last) break;
    }
    if (h.has_checksum) {
        if (n - pos < 4) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        pos += 4;
    }
    return pos;
}

/* ------------------------------------------------------------------ FSE */
typedef struct {
    u8 sym;
    u8 nbits;
    u16 base; /* next state = base + read(nbits) */
} fse_entry;

typedef struct {
    int log;
    fse_entry e[512];
} fse_table;

/* zstd.c:3269-3446 FSE_readNCount_body.  Forward, LSB-first bit reads.
 * Returns bytes consumed or error; fills norm[0..*max_sym]. */
static size_t read_ncount(short *norm, int *max_sym, int *table_log, int max_log, const u8 *src, size_t n) {
    if (n < 1) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    /* forward reads never touch bytes past src+n: work on a zero-padded copy of the (short) header */
    u8 padded[520];
    size_t keep = n < 512 ? n : 512;
    memset(padded, 0, sizeof(padded));
    memcpy(padded, src, keep);
    src = padded;
    i64 bit = 0;
    i64 total_bits = (i64)keep * 8;
    int log = (int)bits_at(src, bit, 4) + 5;
    bit += 4;
    if (log > 15) return ERR(ZLO_ERR_TABLELOG_TOO_LARGE); /* FSE_TABLELOG_ABSOLUTE_MAX */
    if (log > max_log) return ERR(ZLO_ERR_TABLELOG_TOO_LARGE);
    int remaining = (1 << log) + 1;
    int threshold = 1 << log;
    int nbits = log + 1;
    int sym = 0;
    int limit = *max_sym;
    int prev_zero = 0;
    while (remaining > 1 && sym <= limit) {
        if (prev_zero) {
            /* run of zero-probability symbols, 2 bits at a time, 3 == continue */
            for (;;) {
                if (bit + 2 > total_bits) return ERR(ZLO_ERR_CORRUPTION);
                int r = (int)bits_at(src, bit, 2);
                bit += 2;
                sym += r;
                if (r != 3) break;
            }
            if (sym > limit) return ERR(ZLO_ERR_MAXSYMBOL_TOO_SMALL);
            prev_zero = 0;
            /* zeros already in norm[] (caller memset) */
        }
        int max = (2 * threshold - 1) - remaining;
        int count;
        if (bit > total_bits) return ERR(ZLO_ERR_CORRUPTION);
        u32 peek = bits_at(src, bit, nbits);
        if ((int)(peek & (u32)(threshold - 1)) < max) {
            count = (int)(peek & (u32)(threshold - 1));
            bit += nbits - 1;
        } else {
            count = (int)(peek & (u32)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            bit += nbits;
        }
        count--; /* extra accuracy: -1 means "less than one" */
        if (count >= 0) remaining -= count; else remaining += count; /* |count| */
        norm[sym++] = (short)count;
        prev_zero = (count == 0);
        while (remaining < threshold && threshold > 1) { nbits--; threshold >>= 1; }
    }
    if (remaining != 1) return ERR(ZLO_ERR_CORRUPTION);
    if (sym > limit + 1) return ERR(ZLO_ERR_MAXSYMBOL_TOO_SMALL);
    if (bit > total_bits) return ERR(ZLO_ERR_CORRUPTION);
    *max_sym = sym - 1;
    *table_log = log;
    return (size_t)((bit + 7) >> 3);
}

/* zstd.c:3690-3790 FSE_buildDTable_internal / zstd.c:43497-43600 ZSTD_buildFSETable_body:
 * "less than one" symbols go to the top of the table, the rest are spread with the
 * (size/2 + size/8 + 3) step, then each cell gets nbBits/baseline from its rank within
 * the symbol. */
static size_t build_fse_table(fse_table *t, const short *norm, int max_sym, int log) {
    int size = 1 << log;
    int high = size - 1;
    u16 next[256];
    t->log = log;
    for (int s = 0; s <= max_sym; s++) {
        if (norm[s] == -1) { t->e[high--].sym = (u8)s; next[s] = 1; }
        else next[s] = (u16)norm[s];
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s <= max_sym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            t->e[pos].sym = (u8)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return ERR(ZLO_ERR_GENERIC);
    for (int u = 0; u < size; u++) {
        u32 ns = next[t->e[u].sym]++;
        int nb = log - highbit(ns);
        t->e[u].nbits = (u8)nb;
        t->e[u].base = (u16)((ns << nb) - (u32)size);
    }
    return 0;
}

static void build_rle_table(fse_table *t, u8 sym) {
    t->log = 0;
    t->e[0].sym = sym; t->e[0].nbits = 0; t->e[0].base = 0;
}

/* ------------------------------------------------------------------ Huffman */
typedef struct {
    int log;          /* max code length == table log; 0 == no table */
    u8 sym[4096];
    u8 nbits[4096];
} huf_table;

/* zstd.c:3470-3560 HUF_readStats_body: weights either direct 4-bit or FSE-compressed
 * (zstd.c:3808-3870 FSE_decompress_usingDTable_generic, two interleaved states). Then
 * zstd.c:38436-38560 HUF_readDTableX1_wksp: single-symbol decode table. Returns bytes
 * consumed by the tree description. */
static size_t read_huf_table(huf_table *ht, const u8 *src, size_t n) {
    u8 w[256];
    int nsym = 0;
    size_t used;
    memset(w, 0, sizeof(w));
    if (n < 1) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    u8 hb = src[0];
    if (hb >= 128) {
        nsym = hb - 127;
        used = 1 + (size_t)(nsym + 1) / 2;
        if (used > n) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        for (int i = 0; i < nsym; i++) {
            u8 b = src[1 + i / 2];
            w[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
    } else {
        used = 1 + (size_t)hb;
        if (used > n) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        short norm[256];
        int max_sym = 255, log;
        memset(norm, 0, sizeof(norm));
        size_t hs = read_ncount(norm, &max_sym, &log, 6, src + 1, hb);
        if (IS_ERR(hs)) return hs;
        fse_table ft;
        size_t e = build_fse_table(&ft, norm, max_sym, log);
        if (IS_ERR(e)) return e;
        const u8 *bs = src + 1 + hs;
        size_t bn = hb - hs;
        i64 pos = bwd_init(bs, bn);
        if (pos < 0) return ERR(ZLO_ERR_CORRUPTION);
        pos -= log; u32 s1 = bits_at(bs, pos, log);
        pos -= log; u32 s2 = bits_at(bs, pos, log);
        if (pos < 0) return ERR(ZLO_ERR_CORRUPTION);
        for (;;) {
            if (nsym > 253) return ERR(ZLO_ERR_CORRUPTION);
            w[nsym++] = ft.e[s1].sym;
            pos -= ft.e[s1].nbits;
            s1 = ft.e[s1].base + bits_at(bs, pos, ft.e[s1].nbits);
            if (pos < 0) { w[nsym++] = ft.e[s2].sym; break; }
            if (nsym > 253) return ERR(ZLO_ERR_CORRUPTION);
            w[nsym++] = ft.e[s2].sym;
            pos -= ft.e[s2].nbits;
            s2 = ft.e[s2].base + bits_at(bs, pos, ft.e[s2].nbits);
            if (pos < 0) { w[nsym++] = ft.e[s1].sym; break; }
        }
    }
    /* implied last weight */
    u32 total = 0;
    for (int i = 0; i < nsym; i++) {
        if (w[i] > 12) return ERR(ZLO_ERR_CORRUPTION);
        if (w[i]) total += 1u << (w[i] - 1);
    }
    if (total == 0) return ERR(ZLO_ERR_CORRUPTION);
    int log = highbit(total) + 1;
    if (log > 12) return ERR(ZLO_ERR_CORRUPTION);
    u32 rest = (1u << log) - total;
    if (rest & (rest - 1)) return ERR(ZLO_ERR_CORRUPTION); /* must be a power of two */
    w[nsym++] = (u8)(highbit(rest) + 1);
    /* zstd.c:3549: at least two symbols of weight 1 -> rank[1] must be even and >= 2 */
    {
        int r1 = 0;
        for (int i = 0; i < nsym; i++) r1 += (w[i] == 1);
        if (r1 < 2 || (r1 & 1)) return ERR(ZLO_ERR_CORRUPTION);
    }
    /* table fill: ascending weight, ascending symbol; symbol of weight w owns 2^(w-1) cells */
    ht->log = log;
    u32 cell = 0;
    for (int wt = 1; wt <= log; wt++) {
        for (int s = 0; s < nsym; s++) {
            if (w[s] != wt) continue;
            u32 cnt = 1u << (wt - 1);
            for (u32 k = 0; k < cnt; k++) { ht->sym[cell + k] = (u8)s; ht->nbits[cell + k] = (u8)(log + 1 - wt); }
            cell += cnt;
        }
    }
    return used;
}

/* zstd.c:38560-38650 HUF_decompress1X1_usingDTable_internal_body: one backward stream. */
static size_t huf_decode_stream(u8 *dst, size_t nout, const u8 *src, size_t n, const huf_table *ht) {
    i64 pos = bwd_init(src, n);
    if (pos < 0) return ERR(ZLO_ERR_CORRUPTION);
    for (size_t i = 0; i < nout; i++) {
        u32 v = bits_at(src, pos - ht->log, ht->log);
        dst[i] = ht->sym[v];
        pos -= ht->nbits[v];
    }
    if (pos != 0) return ERR(ZLO_ERR_CORRUPTION); /* zstd.c:38640 BIT_endOfDStream */
    return nout;
}

/* ------------------------------------------------------------------ block decode state */
typedef struct {
    huf_table huf;
    int huf_valid;
    fse_table ll, of, ml;
    int ll_valid, of_valid, ml_valid; /* "repeat" allowed */
    u32 rep[3];
    const u8 *dict; /* content bytes that virtually precede the frame output */
    size_t dict_len;
    u8 *lit; /* literals buffer BLOCK_MAX */
} dstate;

static const u32 LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
static const u8 LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3,
                               4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const u32 ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26,
                                27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103,
                                0x203, 0x403, 0x803, 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
static const u8 ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                               0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
/* zstd.c:15470-15512 default distributions */
static const short LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                     2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const short ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                     1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const short OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

/* zstd.c:43146-43351 ZSTD_decodeLiteralsBlock.  Returns bytes consumed; *lit_size out. */
static size_t decode_literals(dstate *st, const u8 *src, size_t n, size_t *lit_size) {
    if (n < 1) return ERR(ZLO_ERR_CORRUPTION);
    int type = src[0] & 3, sf = (src[0] >> 2) & 3;
    if (type < 2) { /* raw / rle */
        size_t hs, regen;
        if (sf == 0 || sf == 2) { hs = 1; regen = src[0] >> 3; }
        else if (sf == 1) { if (n < 2) return ERR(ZLO_ERR_CORRUPTION); hs = 2; regen = rd_le(src, 2) >> 4; }
        else { if (n < 3) return ERR(ZLO_ERR_CORRUPTION); hs = 3; regen = rd_le(src, 3) >> 4; }
        if (regen > BLOCK_MAX) return ERR(ZLO_ERR_CORRUPTION);
        if (type == 0) {
            if (hs + regen > n) return ERR(ZLO_ERR_CORRUPTION);
            memcpy(st->lit, src + hs, regen);
            *lit_size = regen;
            return hs + regen;
        }
        if (hs + 1 > n) return ERR(ZLO_ERR_CORRUPTION);
        memset(st->lit, src[hs], regen);
        *lit_size = regen;
        return hs + 1;
    }
    /* compressed (2) / treeless (3) */
    if (n < 5) return ERR(ZLO_ERR_CORRUPTION); /* zstd.c:43165: srcSize >= MIN_CBLOCK_SIZE+3 for these modes */
    size_t hs, regen, csize;
    int streams = 4;
    u32 h4 = rd_le(src, 4);
    switch (sf) {
    case 0: streams = 1; /* fallthrough */
    case 1: hs = 3; regen = (h4 >> 4) & 0x3FF; csize = (h4 >> 14) & 0x3FF; break;
    case 2: hs = 4; regen = (h4 >> 4) & 0x3FFF; csize = h4 >> 18; break;
    default: hs = 5; regen = (h4 >> 4) & 0x3FFFF; csize = (h4 >> 22) + ((size_t)src[4] << 10); break;
    }
    if (regen > BLOCK_MAX) return ERR(ZLO_ERR_CORRUPTION);
    if (streams == 4 && regen < 6) return ERR(ZLO_ERR_LITERALS_HEADER); /* zstd.c:43216 MIN_LITERALS_FOR_4_STREAMS */
    if (hs + csize > n) return ERR(ZLO_ERR_CORRUPTION);
    const u8 *p = src + hs;
    size_t left = csize;
    if (type == 3) {
        if (!st->huf_valid) return ERR(ZLO_ERR_DICT_CORRUPTED); /* zstd.c:43171 */
    } else {
        size_t used = read_huf_table(&st->huf, p, left);
        if (IS_ERR(used)) return ERR(ZLO_ERR_CORRUPTION);
        p += used; left -= used;
        st->huf_valid = 1;
    }
    if (streams == 1) {
        size_t e = huf_decode_stream(st->lit, regen, p, left, &st->huf);
        if (IS_ERR(e)) return ERR(ZLO_ERR_CORRUPTION);
    } else {
        if (left < 10) return ERR(ZLO_ERR_CORRUPTION); /* zstd.c:38665: jump table + 1 byte per stream */
        size_t s1 = rd_le(p, 2), s2 = rd_le(p + 2, 2), s3 = rd_le(p + 4, 2);
        if (6 + s1 + s2 + s3 > left) return ERR(ZLO_ERR_CORRUPTION);
        size_t s4 = left - 6 - s1 - s2 - s3;
        size_t seg = (regen + 3) / 4;
        if (seg * 3 > regen) return ERR(ZLO_ERR_CORRUPTION);
        const u8 *q = p + 6;
        size_t e;
        e = huf_decode_stream(st->lit, seg, q, s1, &st->huf); if (IS_ERR(e)) return ERR(ZLO_ERR_CORRUPTION);
        e = huf_decode_stream(st->lit + seg, seg, q + s1, s2, &st->huf); if (IS_ERR(e)) return ERR(ZLO_ERR_CORRUPTION);
        e = huf_decode_stream(st->lit + 2 * seg, seg, q + s1 + s2, s3, &st->huf); if (IS_ERR(e)) return ERR(ZLO_ERR_CORRUPTION);
        e = huf_decode_stream(st->lit + 3 * seg, regen - 3 * seg, q + s1 + s2 + s3, s4, &st->huf); if (IS_ERR(e)) return ERR(ZLO_ERR_CORRUPTION);
    }
    *lit_size = regen;
    return hs + csize;
}

/* zstd.c:43640-43700 ZSTD_buildSeqTable: one of predefined / rle / compressed / repeat. */
static size_t read_seq_table(fse_table *t, int *valid, int mode, const u8 *src, size_t n, const short *def_norm,
                             int def_max, int def_log, int max_sym, int max_log) {
    switch (mode) {
    case 0: {
        size_t e = build_fse_table(t, def_norm, def_max, def_log);
        if (IS_ERR(e)) return e;
        *valid = 1;
        return 0;
    }
    case 1:
        if (n < 1) return ERR(ZLO_ERR_SRCSIZE_WRONG);
        if (src[0] > max_sym) return ERR(ZLO_ERR_CORRUPTION);
        build_rle_table(t, src[0]);
        *valid = 1;
        return 1;
    case 2: {
        short norm[64];
        int ms = max_sym, log;
        memset(norm, 0, sizeof(norm));
        size_t hs = read_ncount(norm, &ms, &log, max_log, src, n);
        if (IS_ERR(hs)) return ERR(ZLO_ERR_CORRUPTION);
        size_t e = build_fse_table(t, norm, ms, log);
        if (IS_ERR(e)) return ERR(ZLO_ERR_CORRUPTION);
        *valid = 1;
        return hs;
    }
    default:
        if (!*valid) return ERR(ZLO_ERR_CORRUPTION);
        return 0;
    }
}

typedef struct { u32 ll, ml, off; } seq_t;

/* zstd.c:43707-43786 ZSTD_decodeSeqHeaders + zstd.c:44241-44356 ZSTD_decodeSequence
 * (3 interleaved FSE states in one backward bitstream; offset repcode rules
 * zstd.c:44291-44327).  Writes sequences to seqs (cap >= nbSeq). Returns nbSeq. */
static size_t decode_sequences(dstate *st, const u8 *src, size_t n, seq_t *seqs, size_t cap) {
    if (n < 1) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    size_t pos = 0;
    u32 nseq = src[pos++];
    if (nseq > 0x7F) {
        if (nseq == 0xFF) {
            if (n < 3) return ERR(ZLO_ERR_SRCSIZE_WRONG);
            nseq = rd_le(src + 1, 2) + 0x7F00; pos = 3;
        } else {
            if (n < 2) return ERR(ZLO_ERR_SRCSIZE_WRONG);
            nseq = ((nseq - 0x80) << 8) + src[1]; pos = 2;
        }
    }
    if (nseq == 0) {
        if (pos != n) return ERR(ZLO_ERR_CORRUPTION); /* zstd.c:43733 */
        return 0;
    }
    if (nseq > cap) return ERR(ZLO_ERR_CORRUPTION);
    if (pos + 1 > n) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    u8 modes = src[pos++];
    if (modes & 3) return ERR(ZLO_ERR_CORRUPTION); /* reserved bits, zstd.c:43741 */
    size_t r;
    r = read_seq_table(&st->ll, &st->ll_valid, modes >> 6, src + pos, n - pos, LL_DEFAULT, 35, 6, 35, 9);
    if (IS_ERR(r)) return ERR(ZLO_ERR_CORRUPTION);
    pos += r;
    r = read_seq_table(&st->of, &st->of_valid, (modes >> 4) & 3, src + pos, n - pos, OF_DEFAULT, 28, 5, 31, 8);
    if (IS_ERR(r)) return ERR(ZLO_ERR_CORRUPTION);
    pos += r;
    r = read_seq_table(&st->ml, &st->ml_valid, (modes >> 2) & 3, src + pos, n - pos, ML_DEFAULT, 52, 6, 52, 9);
    if (IS_ERR(r)) return ERR(ZLO_ERR_CORRUPTION);
    pos += r;

    const u8 *bs = src + pos;
    size_t bn = n - pos;
    i64 bp = bwd_init(bs, bn);
    if (bp < 0) return ERR(ZLO_ERR_CORRUPTION);
    bp -= st->ll.log; u32 sl = bits_at(bs, bp, st->ll.log);
    bp -= st->of.log; u32 so = bits_at(bs, bp, st->of.log);
    bp -= st->ml.log; u32 sm = bits_at(bs, bp, st->ml.log);
    if (bp < 0) return ERR(ZLO_ERR_CORRUPTION);
    for (u32 i = 0; i < nseq; i++) {
        u32 llc = st->ll.e[sl].sym, ofc = st->of.e[so].sym, mlc = st->ml.e[sm].sym;
        if (llc > 35 || mlc > 52 || ofc > 31) return ERR(ZLO_ERR_CORRUPTION);
        bp -= ofc; u32 ofv = (1u << ofc) + bits_at(bs, bp, (int)ofc);
        bp -= ML_BITS[mlc]; u32 ml = ML_BASE[mlc] + bits_at(bs, bp, ML_BITS[mlc]);
        bp -= LL_BITS[llc]; u32 ll = LL_BASE[llc] + bits_at(bs, bp, LL_BITS[llc]);
        if (bp < 0) return ERR(ZLO_ERR_CORRUPTION);
        /* repcode resolution */
        u32 off;
        if (ofv > 3) {
            off = ofv - 3;
            st->rep[2] = st->rep[1]; st->rep[1] = st->rep[0]; st->rep[0] = off;
        } else {
            u32 idx = ofv - 1 + (ll == 0); /* 0..3 */
            if (idx == 0) off = st->rep[0];
            else {
                off = (idx == 3) ? st->rep[0] - 1 : st->rep[idx];
                if (off == 0) return ERR(ZLO_ERR_CORRUPTION); /* zstd.c:44320 comment: 0 is not valid */
                if (idx != 1) st->rep[2] = st->rep[1];
                st->rep[1] = st->rep[0];
                st->rep[0] = off;
            }
        }
        seqs[i].ll = ll; seqs[i].ml = ml; seqs[i].off = off;
        if (i + 1 < nseq) {
            bp -= st->ll.e[sl].nbits; sl = st->ll.e[sl].base + bits_at(bs, bp, st->ll.e[sl].nbits);
            bp -= st->ml.e[sm].nbits; sm = st->ml.e[sm].base + bits_at(bs, bp, st->ml.e[sm].nbits);
            bp -= st->of.e[so].nbits; so = st->of.e[so].base + bits_at(bs, bp, st->of.e[so].nbits);
            if (bp < 0) return ERR(ZLO_ERR_CORRUPTION);
        }
    }
    if (bp != 0) return ERR(ZLO_ERR_CORRUPTION); /* zstd.c:44690 BIT_endOfDStream */
    return nseq;
}

/* zstd.c:45084-45190 ZSTD_decompressBlock_internal + zstd.c:44013-44104 ZSTD_execSequence.
 * out_base is the frame's first output byte; op the current position. Returns bytes
 * regenerated. */
static size_t decode_block(dstate *st, u8 *out_base, size_t op, size_t cap, const u8 *src, size_t n, seq_t *seqs) {
    size_t lit_size = 0;
    if (n > BLOCK_MAX) return ERR(ZLO_ERR_SRCSIZE_WRONG); /* zstd.c:45107 */
    size_t lc = decode_literals(st, src, n, &lit_size);
    if (IS_ERR(lc)) return lc;
    size_t nseq = decode_sequences(st, src + lc, n - lc, seqs, BLOCK_MAX / 3 + 16);
    if (IS_ERR(nseq)) return nseq;
    size_t lp = 0, start = op;
    for (size_t i = 0; i < nseq; i++) {
        u32 ll = seqs[i].ll, ml = seqs[i].ml, off = seqs[i].off;
        if (ll > lit_size - lp) return ERR(ZLO_ERR_CORRUPTION);
        if ((size_t)ll + ml > cap - op) return ERR(ZLO_ERR_DSTSIZE_TOO_SMALL);
        memcpy(out_base + op, st->lit + lp, ll);
        op += ll; lp += ll;
        if (off > op + st->dict_len) return ERR(ZLO_ERR_CORRUPTION);
        for (u32 k = 0; k < ml; k++) {
            i64 s = (i64)op - (i64)off;
            out_base[op] = (s >= 0) ? out_base[s] : st->dict[(i64)st->dict_len + s];
            op++;
        }
    }
    size_t rest = lit_size - lp;
    if (rest > cap - op) return ERR(ZLO_ERR_DSTSIZE_TOO_SMALL);
    memcpy(out_base + op, st->lit + lp, rest);
    op += rest;
    if (op - start > BLOCK_MAX) return ERR(ZLO_ERR_CORRUPTION);
    return op - start;
}

/* ------------------------------------------------------------------ dictionary */
unsigned zlo_dict_id(const void *dict, size_t n) {
    if (n < 8 || rd_le((const u8 *)dict, 4) != ZSTD_MAGIC_DICT) return 0;
    return rd_le((const u8 *)dict + 4, 4);
}

/* zstd.c:42053-42150 ZSTD_loadDEntropy + zstd.c:42160-42190 ZSTD_decompress_insertDictionary. */
static size_t load_dict(dstate *st, const u8 *dict, size_t n) {
    st->dict = dict; st->dict_len = n;
    if (n < 8 || rd_le(dict, 4) != ZSTD_MAGIC_DICT) return 0; /* raw content */
    size_t pos = 8;
    size_t r = read_huf_table(&st->huf, dict + pos, n - pos);
    if (IS_ERR(r)) return ERR(ZLO_ERR_DICT_CORRUPTED);
    pos += r; st->huf_valid = 1;
    short norm[64]; int ms, log;
    memset(norm, 0, sizeof(norm)); ms = 31;
    r = read_ncount(norm, &ms, &log, 8, dict + pos, n - pos);
    if (IS_ERR(r)) return ERR(ZLO_ERR_DICT_CORRUPTED);
    if (IS_ERR(build_fse_table(&st->of, norm, ms, log))) return ERR(ZLO_ERR_DICT_CORRUPTED);
    pos += r; st->of_valid = 1;
    memset(norm, 0, sizeof(norm)); ms = 52;
    r = read_ncount(norm, &ms, &log, 9, dict + pos, n - pos);
    if (IS_ERR(r)) return ERR(ZLO_ERR_DICT_CORRUPTED);
    if (IS_ERR(build_fse_table(&st->ml, norm, ms, log))) return ERR(ZLO_ERR_DICT_CORRUPTED);
    pos += r; st->ml_valid = 1;
    memset(norm, 0, sizeof(norm)); ms = 35;
    r = read_ncount(norm, &ms, &log, 9, dict + pos, n - pos);
    if (IS_ERR(r)) return ERR(ZLO_ERR_DICT_CORRUPTED);
    if (IS_ERR(build_fse_table(&st->ll, norm, ms, log))) return ERR(ZLO_ERR_DICT_CORRUPTED);
    pos += r; st->ll_valid = 1;
    if (pos + 12 > n) return ERR(ZLO_ERR_DICT_CORRUPTED);
    size_t content = n - (pos + 12);
    for (int i = 0; i < 3; i++) {
        u32 rep = rd_le(dict + pos + 4 * i, 4);
        if (rep == 0 || rep > content) return ERR(ZLO_ERR_DICT_CORRUPTED);
        st->rep[i] = rep;
    }
    pos += 12;
    st->dict = dict + pos; st->dict_len = content;
    return 0;
}

/* ------------------------------------------------------------------ frame / multi-frame */
/* zstd.c:41554-41660 ZSTD_decompressFrame. *consumed = compressed bytes of this frame. */
static size_t decode_frame(u8 *dst, size_t cap, const u8 *src, size_t n, const u8 *dict, size_t dict_size,
                           int ignore_checksum, size_t *consumed, u8 *lit, seq_t *seqs) {
    frame_hdr h;
    size_t e = parse_frame_header(&h, src, n);
    if (IS_ERR(e)) return e;
    dstate *st = (dstate *)calloc(1, sizeof(dstate));
    if (!st) return ERR(64);
    st->rep[0] = 1; st->rep[1] = 4; st->rep[2] = 8;
    st->lit = lit;
    size_t ret, pos, op;
    if (dict && dict_size) {
        e = load_dict(st, dict, dict_size);
        if (IS_ERR(e)) { ret = e; goto done; }
        u32 did = zlo_dict_id(dict, dict_size);
        if (h.dict_id && did != h.dict_id) { ret = ERR(ZLO_ERR_DICT_WRONG); goto done; } /* zstd.c:41318 */
    } else if (h.dict_id) {
        /* zstd.c:41318: dctx->dictID (0) != fParams.dictID */
        ret = ERR(ZLO_ERR_DICT_WRONG); goto done;
    }
    pos = h.header_size; op = 0;
    for (;;) {
        if (n - pos < 3) { ret = ERR(ZLO_ERR_SRCSIZE_WRONG); goto done; }
        u32 bh = rd_le(src + pos, 3);
        int last = bh & 1, type = (bh >> 1) & 3;
        u32 bsize = bh >> 3;
        pos += 3;
        if (type == 3) { ret = ERR(ZLO_ERR_CORRUPTION); goto done; }
        size_t csize = (type == 1) ? 1 : bsize;
        if (n - pos < csize) { ret = ERR(ZLO_ERR_SRCSIZE_WRONG); goto done; }
        if (type == 0) {
            if (bsize > cap - op) { ret = ERR(ZLO_ERR_DSTSIZE_TOO_SMALL); goto done; }
            memcpy(dst + op, src + pos, bsize); op += bsize;
        } else if (type == 1) {
            if (bsize > cap - op) { ret = ERR(ZLO_ERR_DSTSIZE_TOO_SMALL); goto done; }
            memset(dst + op, src[pos], bsize); op += bsize;
        } else {
            size_t r = decode_block(st, dst, op, cap, src + pos, csize, seqs);
            if (IS_ERR(r)) { ret = r; goto done; }
            op += r;
        }
        pos += csize;
        if (last) break;
    }
    if (h.content_size != ZLO_CONTENTSIZE_UNKNOWN && h.content_size != op) { ret = ERR(ZLO_ERR_CORRUPTION); goto done; }
    if (h.has_checksum) {
        if (n - pos < 4) { ret = ERR(ZLO_ERR_CHECKSUM_WRONG); goto done; }
        if (!ignore_checksum) {
            u32 want = rd_le(src + pos, 4);
            u32 got = (u32)zlo_xxh64(dst, op, 0);
            if (want != got) { ret = ERR(ZLO_ERR_CHECKSUM_WRONG); goto done; }
        }
        pos += 4;
    }
    *consumed = pos;
    ret = op;
done:
    free(st);
    return ret;
}

/* zstd.c:41671-41767 ZSTD_decompressMultiFrame: concatenated frames, skippable frames
 * skipped; trailing garbage shorter than a frame start is srcSize_wrong. */
size_t zlo_decompress(void *dst_, size_t cap, const void *src_, size_t n, const void *dict, size_t dict_size,
                      int ignore_checksum) {
    u8 *dst = (u8 *)dst_;
    const u8 *src = (const u8 *)src_;
    u8 *lit = (u8 *)malloc(BLOCK_MAX + 64);
    seq_t *seqs = (seq_t *)malloc(sizeof(seq_t) * (BLOCK_MAX / 3 + 16));
    size_t total = 0, ret;
    int more_than_one = 0;
    if (!lit || !seqs) { ret = ERR(64); goto out; }
    while (n >= 5) { /* ZSTD_startingInputLength */
        u32 magic = rd_le(src, 4);
        if ((magic & 0xFFFFFFF0u) == ZSTD_MAGIC_SKIPPABLE) {
            size_t sk = zlo_find_frame_compressed_size(src, n);
            if (IS_ERR(sk)) { ret = sk; goto out; }
            src += sk; n -= sk;
            continue;
        }
        size_t consumed = 0;
        size_t r = decode_frame(dst + total, cap - total, src, n, (const u8 *)dict, dict_size, ignore_checksum,
                                &consumed, lit, seqs);
        if (IS_ERR(r)) {
            /* zstd.c:41752: prefix_unknown on a 2nd+ frame is reported as srcSize_wrong */
            if (more_than_one && r == ERR(ZLO_ERR_PREFIX_UNKNOWN)) r = ERR(ZLO_ERR_SRCSIZE_WRONG);
            ret = r; goto out;
        }
        total += r;
        src += consumed; n -= consumed;
        more_than_one = 1;
    }
    if (n) { ret = ERR(ZLO_ERR_SRCSIZE_WRONG); goto out; }
    ret = total;
out:
    free(lit); free(seqs);
    return ret;
}

size_t zlo_debug_first_block_sequences(const void *src_, size_t n, uint32_t *triples, size_t cap, size_t *lit_size) {
    const u8 *src = (const u8 *)src_;
    frame_hdr h;
    size_t e = parse_frame_header(&h, src, n);
    if (IS_ERR(e)) return e;
    size_t pos = h.header_size;
    if (n - pos < 3) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    u32 bh = rd_le(src + pos, 3);
    pos += 3;
    if (((bh >> 1) & 3) != 2) return ERR(ZLO_ERR_GENERIC);
    u32 bsize = bh >> 3;
    if (n - pos < bsize) return ERR(ZLO_ERR_SRCSIZE_WRONG);
    dstate *st = (dstate *)calloc(1, sizeof(dstate));
    u8 *lit = (u8 *)malloc(BLOCK_MAX + 64);
    seq_t *seqs = (seq_t *)malloc(sizeof(seq_t) * (BLOCK_MAX / 3 + 16));
    st->rep[0] = 1; st->rep[1] = 4; st->rep[2] = 8; st->lit = lit;
    size_t ret;
    size_t lc = decode_literals(st, src + pos, bsize, lit_size);
    if (IS_ERR(lc)) { ret = lc; goto out; }
    ret = decode_sequences(st, src + pos + lc, bsize - lc, seqs, BLOCK_MAX / 3 + 16);
    if (!IS_ERR(ret)) {
        for (size_t i = 0; i < ret && i < cap; i++) {
            triples[3 * i] = seqs[i].ll; triples[3 * i + 1] = seqs[i].ml; triples[3 * i + 2] = seqs[i].off;
        }
    }
out:
    free(st); free(lit); free(seqs);
    return ret;
}
