/* A C99 consumer of include/zl_hip.h, written the way the reference's C glue calls libzstd
 * (src/raw-file.c:52-111 compress, :150-192 decompress, src/raw-file-out.c:74-123 and
 * src/raw-file-in.c:87-103 streaming) with the zl_ twins in place of the ZSTD_ calls.  It proves that
 * the header is plain C and that the library links and behaves from C; tests/test_abi.py builds it with
 * gcc and runs it.  Without a HIP device the compute calls must fail with an error code, not crash. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zl_hip.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL line %d: %s\n", __LINE__, msg); return 1; } } while (0)

int main(void) {
    const size_t n = 3u << 20;
    unsigned char *src = (unsigned char *)malloc(n), *back = (unsigned char *)malloc(n);
    size_t i, bound, csize, fsize, total, pos;
    unsigned char *dst;
    zl_cctx *cctx;
    zl_dctx *dctx;
    int value = 0;
    CHECK(src && back, "malloc");
    for (i = 0; i < n; i++) src[i] = (unsigned char)("the quick brown fox "[i % 20] + (i / 7919) % 3);

    printf("%s, %d device(s)\n", zl_versionString(), zl_deviceCount());
    bound = zl_compressBound(n);                                         /* raw-file.c:52 */
    CHECK(bound >= n, "bound");
    dst = (unsigned char *)malloc(bound);
    CHECK(dst, "malloc");
    cctx = zl_createCCtx();                                              /* cctx.c:222 */
    dctx = zl_createDCtx();                                              /* dctx.c:119 */
    CHECK(cctx && dctx, "contexts");
    CHECK(!zl_isError(zl_CCtx_setParameter(cctx, 100 /* ZSTD_c_compressionLevel */, 3)), "level");
    CHECK(zl_isError(zl_CCtx_setParameter(cctx, 100, 1000)), "level out of range is an error");
    CHECK(strcmp(zl_getErrorName((size_t)0 - 20), "Data corruption detected") == 0, "error string");

    csize = zl_compress2(cctx, dst, bound, src, n);                      /* raw-file.c:74 */
    if (zl_deviceCount() == 0) {
        CHECK(zl_isError(csize), "no device: compress must report an error");
        CHECK(zl_isError(zl_decompressDCtx(dctx, back, n, "\x28\xb5\x2f\xfd\x20\x00\x01\x00\x00", 9)), "no device: decompress must report an error");
        printf("no HIP device: compute entries report '%s'\n", zl_getErrorName(csize));
    } else {
        zl_inBuffer in;
        zl_outBuffer out;
        size_t rem, spos = 0, csize2;
        unsigned long long fcs;
        unsigned char *stream = (unsigned char *)malloc(bound), *dst2 = (unsigned char *)malloc(bound);
        CHECK(stream && dst2, "malloc");
        CHECK(!zl_isError(csize), zl_getErrorName(csize));
        /* exactly the reference glue's read side, first-frame helpers only (raw-file.c:150-189): ONE frame holds all 3 MiB */
        fsize = zl_findFrameCompressedSize(dst, csize);                  /* raw-file.c:150 */
        CHECK(!zl_isError(fsize) && fsize == csize, "zl_compress2 writes one frame");
        fcs = zl_getFrameContentSize(dst, fsize);                        /* raw-file.c:155 */
        CHECK(fcs == n, "the frame records the whole content size");
        total = zl_decompressDCtx(dctx, back, (size_t)fcs, dst, fsize);  /* raw-file.c:189 */
        CHECK(total == n && memcmp(src, back, n) == 0, "one-shot round trip through the first-frame helpers");
        /* a checksum context writes ONE frame too (zstd.c:27733: one XXH64 over the whole content, four bytes behind the last
         * block); the independent-frames layout, on request, is what the whole-buffer twins size */
        CHECK(!zl_isError(zl_CCtx_setParameter(cctx, 201 /* ZSTD_c_checksumFlag */, 1)), "checksum flag");
        CHECK(!zl_isError(zl_CCtx_getParameter(cctx, 201, &value)) && value == 1, "get parameter");
        csize2 = zl_compress2(cctx, dst2, bound, src, n);
        CHECK(!zl_isError(csize2), zl_getErrorName(csize2));
        CHECK(csize2 == csize + 4 && zl_findFrameCompressedSize(dst2, csize2) == csize2 && zl_getFrameContentSize(dst2, csize2) == n, "checksum: still one frame, four bytes longer");
        memset(back, 0, n);
        total = zl_decompressDCtx(dctx, back, n, dst2, csize2);
        CHECK(total == n && memcmp(src, back, n) == 0, "round trip of the checksummed frame");
        dst2[csize2 - 1] ^= 1;
        total = zl_decompressDCtx(dctx, back, n, dst2, csize2);
        CHECK(zl_isError(total) && strcmp(zl_getErrorName(total), "Restored data doesn't match checksum") == 0, "a damaged checksum is reported (tests/testthat/test-checksums.R:24-27)");
        CHECK(!zl_isError(zl_CCtx_setParameter(cctx, 9003 /* ZL_c_frameLayout */, 1 /* ZL_layout_frames */)), "frame layout");
        csize2 = zl_compress2(cctx, dst2, bound, src, n);
        CHECK(!zl_isError(csize2), zl_getErrorName(csize2));
        CHECK(zl_getFrameContentSize(dst2, csize2) == 131072, "independent frames on request: one per 128 KiB chunk");
        CHECK(zl_findDecompressedSize(dst2, csize2) == n && zl_findAllFramesCompressedSize(dst2, csize2) == csize2, "whole-buffer size twins");
        memset(back, 0, n);
        total = zl_decompressDCtx(dctx, back, n, dst2, csize2);
        CHECK(total == n && memcmp(src, back, n) == 0, "round trip of concatenated checksummed frames");
        CHECK(!zl_isError(zl_CCtx_setParameter(cctx, 9003, 0)), "frame layout back to the default");
        CHECK(!zl_isError(zl_CCtx_setParameter(cctx, 201, 0)), "checksum flag off");
        free(dst2);
        /* streaming out (raw-file-out.c:74-123): pledge, whole buffer as input, output in pieces */
        CHECK(!zl_isError(zl_CCtx_setPledgedSrcSize(cctx, n)), "pledge");
        in.src = src; in.size = n; in.pos = 0;
        do {
            out.dst = stream + spos; out.size = 100000 < bound - spos ? 100000 : bound - spos; out.pos = 0;
            rem = zl_compressStream2(cctx, &out, &in, ZL_e_end);
            CHECK(!zl_isError(rem), zl_getErrorName(rem));
            spos += out.pos;
        } while (rem != 0);
        CHECK(spos == csize && memcmp(stream, dst, csize) == 0, "the stream equals the one-shot bytes");
        /* streaming in (raw-file-in.c:87-103): one output buffer, input in pieces */
        memset(back, 0, n);
        out.dst = back; out.size = n; out.pos = 0;
        for (pos = 0; pos < spos; pos += 131702) {
            in.src = stream + pos; in.size = spos - pos < 131702 ? spos - pos : 131702; in.pos = 0;
            while (in.pos < in.size) CHECK(!zl_isError(zl_decompressStream(dctx, &out, &in)), "stream decode");
        }
        CHECK(out.pos == n && memcmp(src, back, n) == 0, "streaming round trip");
        free(stream);
        printf("round trips ok: %lu -> %lu bytes\n", (unsigned long)n, (unsigned long)csize);
    }
    CHECK(zl_freeCCtx(cctx) == 0 && zl_freeDCtx(dctx) == 0, "free");     /* cctx.c:191 finalizer */
    free(src); free(back); free(dst);
    printf("ok\n");
    return 0;
}
