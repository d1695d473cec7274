// zl_kernels.h -- launch-argument structs and launcher prototypes shared between the kernel
// translation units and the C-ABI host code (zl_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "zl_format.h"

#define ZL_LIT_STRIDE (ZL_BLOCK_MAX + 256)
#define ZL_SEQ_CAP 43776  // >= ZL_MAX_SEQ_PER_BLOCK, multiple of 64

struct zl_decode_args {
    const zl_frame_desc* frames;
    u32 nframes;
    const u8* src;
    u8* dst;
    u64* results;      // per frame: regenerated size or (size_t)-code
    u32* counter;      // work-queue head (zeroed before launch)
    u8* lit_scratch;   // per slot ZL_LIT_STRIDE bytes
    u32* seq_scratch;  // per slot 3 * ZL_SEQ_CAP words
    const u8* dict_content;  // raw content that virtually precedes every frame (may be null)
    u32 dict_len;
    u32 dict_id;
    const void* dict_tables;  // zl_dict_tables* or null
    u32 ignore_checksum;
};

// digested dictionary entropy (zstd.c:42053 ZSTD_loadDEntropy), produced by
// zl_digest_dict_kernel and copied into LDS at the start of every frame
struct zl_dict_tables {
    u64 fse_ll[512], fse_ml[512], fse_of[256];
    u16 huf[4096];
    u32 ll_log, ml_log, of_log, huf_log;
    u32 rep[3];
    u32 content_off;  // first content byte inside the dictionary buffer
    u32 status;       // 0 ok, else error code
};

extern "C" hipError_t zl_launch_decode(const zl_decode_args* a, u32 nslots, hipStream_t stream);
extern "C" hipError_t zl_launch_digest_dict(const u8* d_dict, u32 n, void* d_tables, hipStream_t stream);
