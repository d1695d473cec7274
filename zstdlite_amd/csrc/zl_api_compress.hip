// zl_api_compress.hip -- compression half of the C-ABI (include/zl_hip.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "../../include/zl_hip.h"
#include "zl_format.h"

struct zl_cctx_s {
    int level = 3;
    int checksum = 0;
    int nb_workers = 0;
    std::vector<u8> dict;
};

extern "C" {

size_t zl_compressBound(size_t n) {
    // per-chunk frame: <= 13 header + 3 block header + chunk + 4 checksum; keep >= ZSTD_COMPRESSBOUND
    const size_t chunks = n / ZL_BLOCK_MAX + 1;
    const size_t zb = n + (n >> 8) + (n < ZL_BLOCK_MAX ? ((ZL_BLOCK_MAX - n) >> 11) : 0);
    return zb + chunks * 24;
}

zl_cctx* zl_createCCtx(void) { return new (std::nothrow) zl_cctx_s(); }
size_t zl_freeCCtx(zl_cctx* c) { delete c; return 0; }
size_t zl_CCtx_setParameter(zl_cctx* c, int param, int value) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    switch (param) {
    case ZL_c_compressionLevel: c->level = value; return 0;
    case ZL_c_checksumFlag: c->checksum = value ? 1 : 0; return 0;
    case ZL_c_nbWorkers: c->nb_workers = value; return 0;
    case ZL_c_stableInBuffer: case ZL_c_stableOutBuffer: return 0;
    default: return ZL_ERR(ZL_E_PARAM_UNSUPPORTED);
    }
}
size_t zl_CCtx_getParameter(const zl_cctx* c, int param, int* value) {
    if (!c || !value) return ZL_ERR(ZL_E_GENERIC);
    switch (param) {
    case ZL_c_compressionLevel: *value = c->level; return 0;
    case ZL_c_checksumFlag: *value = c->checksum; return 0;
    case ZL_c_nbWorkers: *value = c->nb_workers; return 0;
    default: return ZL_ERR(ZL_E_PARAM_UNSUPPORTED);
    }
}
size_t zl_CCtx_loadDictionary(zl_cctx* c, const void* dict, size_t n) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    c->dict.clear();
    if (dict && n) c->dict.assign((const u8*)dict, (const u8*)dict + n);
    return 0;
}
size_t zl_compress2(zl_cctx*, void*, size_t, const void*, size_t) { return ZL_ERR(ZL_E_GENERIC); }
size_t zl_compress_batch(zl_cctx*, size_t, void* const*, const size_t*, const void* const*, const size_t*, size_t*) { return ZL_ERR(ZL_E_GENERIC); }
size_t zl_compress_device_nframes(size_t n) { return n / ZL_BLOCK_MAX + ((n % ZL_BLOCK_MAX) || n == 0 ? 1 : 0); }
size_t zl_compress_device_async(zl_cctx*, void*, size_t, const void*, size_t, void*) { return ZL_ERR(ZL_E_GENERIC); }
size_t zl_compress_device_finish(zl_cctx*, size_t*) { return ZL_ERR(ZL_E_GENERIC); }

}  // extern "C"
