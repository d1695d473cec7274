// zl_decode3.h -- host entry of the large-frame pipeline (zl_decode3.hip), used by the C-ABI (zl_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "zl_format.h"
#include "zl_kernels.h"

typedef struct { u32 src_off; u32 hdr; } zl3_hblk;  // one block of a frame: where its content starts (behind the 3-byte header, relative to the frame) and the header's 24 bits
typedef struct {
    const zl3_hblk* host_blocks;  // the frame's blocks as the host walked them (zl_findFrameCompressedSize does that walk anyway), or null
    u32 nblocks, end_pos;         // their number; where the last block ends
    const u64* d_chunk_off;       // device: nchunks + 1 offsets, relative to the frame, of the runs of blocks this library's one-frame layout wrote per chunk, or null
    u32 nchunks;
    u32 headerless;               // the bytes are a run of a frame's blocks without its header (one device's shard of a frame of this library, section-aligned): the content size is fd->dst_cap, no checksum
    u32 not_last;                 // ... and the run does not hold the frame's last block
} zl3_hint;
typedef struct zl3_state_s zl3_state;  // grow-only device buffers, owned by a decompression context
extern "C" zl3_state* zl3_create(void);
extern "C" void zl3_free(zl3_state* s);
// One frame of more than ZL2_MAXB blocks with a recorded content size.  Returns 0 when the frame's
// result word (size or error code) has been written to *d_result on the device, ZL_ERR(ZL_E_FALLBACK)
// when the general kernel has to take the frame, another error code when the launch itself failed.
// Synchronises the stream several times.
extern "C" size_t zl3_decode_frame(zl3_state* s, const zl_frame_desc* fd, const u8* d_src, u8* d_dst, u64* d_result, int ignore_checksum, hipStream_t st, const zl2_side* side, const zl3_hint* hint);
