// zl_decode3.h -- host entry of the large-frame pipeline (zl_decode3.hip), used by the C-ABI (zl_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "zl_format.h"
#include "zl_kernels.h"

typedef struct zl3_state_s zl3_state;  // grow-only device buffers, owned by a decompression context
extern "C" zl3_state* zl3_create(void);
extern "C" void zl3_free(zl3_state* s);
// One frame of more than ZL2_MAXB blocks with a recorded content size.  Returns 0 when the frame's
// result word (size or error code) has been written to *d_result on the device, ZL_ERR(ZL_E_FALLBACK)
// when the general kernel has to take the frame, another error code when the launch itself failed.
// Synchronises the stream several times.
extern "C" size_t zl3_decode_frame(zl3_state* s, const zl_frame_desc* fd, const u8* d_src, u8* d_dst, u64* d_result, int ignore_checksum, hipStream_t st, const zl2_side* side);
