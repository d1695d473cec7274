/*
 * zl_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the zstd one-shot decode path that the reference
 * package reaches through ZSTD_decompressDCtx (src/raw-file.c:189,
 * src/serialize.c:197 -> zstd.c:41798), plus the frame-size helpers the glue
 * calls first (src/raw-file.c:150,155).  It exists so that tests/, smoke() and
 * bench.py's cpu_baseline leg can check the HIP path; nothing in the product
 * path (zstdlite_amd/) may include, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   - the reference's one binary known-answer fixture (man/figures/data.json.zst,
 *     plaintext from README.md:199-223), committed under tests/golden/,
 *   - golden frames produced by the reference libzstd itself (oracle/_ref,
 *     built from /root/reference/src/zstd/zstd.c by oracle/Makefile) with
 *     SHA-256 of the plaintext, committed under tests/golden/,
 *   - and, when oracle/_ref is present, live differential runs.
 *
 * All file:line citations are relative to /root/reference/ ("zstd.c" is
 * src/zstd/zstd.c).
 */
#ifndef ZL_ORACLE_H
#define ZL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error convention identical to libzstd: (size_t)-code, code from the
 * ZSTD_ErrorCode enum (zstd.c:1463-1499); isError == code > (size_t)-120
 * (zstd.c:1548). */
#define ZLO_ERR_GENERIC 1
#define ZLO_ERR_PREFIX_UNKNOWN 10
#define ZLO_ERR_FRAMEPARAM_UNSUPPORTED 14
#define ZLO_ERR_WINDOW_TOO_LARGE 16
#define ZLO_ERR_CORRUPTION 20
#define ZLO_ERR_CHECKSUM_WRONG 22
#define ZLO_ERR_LITERALS_HEADER 24
#define ZLO_ERR_DICT_CORRUPTED 30
#define ZLO_ERR_DICT_WRONG 32
#define ZLO_ERR_TABLELOG_TOO_LARGE 44
#define ZLO_ERR_MAXSYMBOL_TOO_LARGE 46
#define ZLO_ERR_MAXSYMBOL_TOO_SMALL 48
#define ZLO_ERR_DSTSIZE_TOO_SMALL 70
#define ZLO_ERR_SRCSIZE_WRONG 72
#define ZLO_ERR_MAXCODE 120

#define ZLO_CONTENTSIZE_UNKNOWN (0ULL - 1)
#define ZLO_CONTENTSIZE_ERROR (0ULL - 2)

unsigned zlo_is_error(size_t code);
const char *zlo_error_name(size_t code);

/* zstd.h:231 macro / zstd.c:22583 */
size_t zlo_compress_bound(size_t src_size);

/* zstd.c:41410 (ZSTD_findFrameCompressedSize) via :41335 */
size_t zlo_find_frame_compressed_size(const void *src, size_t src_size);

/* zstd.c:41170 (ZSTD_getFrameContentSize) */
unsigned long long zlo_get_frame_content_size(const void *src, size_t src_size);

/* zstd.c:41798 ZSTD_decompressDCtx -> :41671 ZSTD_decompressMultiFrame.
 * dict may be NULL (dict_size 0).  ignore_checksum mirrors
 * ZSTD_d_forceIgnoreChecksum (src/dctx.c:160-167). */
size_t zlo_decompress(void *dst, size_t dst_cap, const void *src, size_t src_size,
                      const void *dict, size_t dict_size, int ignore_checksum);

/* zstd.c:11708-11800 XXH64 (content checksum = low 32 bits, seed 0). */
uint64_t zlo_xxh64(const void *data, size_t len, uint64_t seed);

/* dictionary id of a formatted dictionary, 0 if raw-content (src/dictionaries.c:23-58
 * -> ZSTD_getDictID_fromDict). */
unsigned zlo_dict_id(const void *dict, size_t dict_size);
/* dictionary id recorded in a frame header (0 if none / not a frame). */
unsigned zlo_frame_dict_id(const void *src, size_t src_size);

/* Debug/inspection hook for kernel bring-up: decode just the entropy stage of the
 * first compressed block of the first frame; returns number of sequences, fills up to
 * cap entries of (ll, ml, off) triples (offset after repcode resolution).  Returns
 * (size_t)-code on error. */
size_t zlo_debug_first_block_sequences(const void *src, size_t src_size, uint32_t *triples,
                                       size_t cap, size_t *lit_size);

#ifdef __cplusplus
}
#endif
#endif
