// zl_kernels.h -- launch-argument structs and launcher prototypes shared between the kernel
// translation units and the C-ABI host code (zl_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "zl_format.h"

#ifndef ZL_MATCH_PRIO
#define ZL_MATCH_PRIO 0  // wave priority (s_setprio) of the match kernel's most expensive chunk class (zl_encode.hip zl_match_auto_kernel); 0: none.  Measured, round 6, 1 GiB mix, same bytes: 3.31 - 3.32 ms without, 3.30 - 3.37 at 2, 3.34 at 3 -- the stragglers of the launch's second round do not wait for issue slots the cheap chunks take
#endif
#ifndef ZL_QUAD_SPLIT
#define ZL_QUAD_SPLIT 1  // the quad chains of single-block launches on two waves (zl_decode2.hip quad_produce / quad_consume); 0: one wave, as rounds 1-4
#endif
#define ZL_LIT_STRIDE (ZL_BLOCK_MAX + 256)
#define ZL_SEQ_CAP 43776  // >= ZL_MAX_SEQ_PER_BLOCK, multiple of 64

// ---- lane-parallel pipeline for frames of at most one block size (zl_decode2.hip) ----------
#define ZL2_MAXB 8          // blocks per frame handled there; more -> general kernel
#define ZL2_WG_FRAMES (64 / ZL2_MAXB)  // frames per entropy workgroup: one sequence lane per (frame, block) fills a wave
#define ZL_E_FALLBACK 118   // internal: "route this frame to the general kernel", never returned to callers
#define ZL_E_FALLBACK_WALK 117  // internal: the caller's chunk offsets do not describe this frame: walk its blocks instead
#define ZL_E_DONE 119       // internal: the frame was decoded by the large-frame pipeline (zl_decode3.hip) before this launch

struct zl2_frame_plan {     // per frame, computed on the host from the frame table
    u64 lit_off;            // byte offset of the frame's literal scratch
    u64 seq_off;            // index of the frame's first 16-byte sequence record
    u32 lit_cap;            // bytes
    u32 seq_cap;            // sequences
    u32 route;              // 1 = lane-parallel pipeline, 0 = general kernel, 3 = already decoded (large-frame pipeline)
    u32 pad;
};
struct zl2_blk_rec {        // per (frame, block), written by zl2_index_kernel
    u32 type, size, src_off;
    u32 lit_mode;           // 0 raw bytes inside the block, 1 RLE, 2 Huffman with table, 3 Huffman reusing the table
    u32 lit_size;
    u32 lit_a;              // raw: offset of the bytes; RLE: the byte; Huffman: header size
    u32 lit_b;              // Huffman: compressed size (tree description + streams)
    u32 seq_off, seq_len;   // sequences section after the sequence-count field
    u32 lit_dst, seq_dst;   // where this block's Huffman literals / sequences go in the frame's scratch
    u32 nseq, streams;
    u32 seq_modes;          // the sequences section's table-mode byte (0x100: missing)
    u32 pad[2];
};
struct zl2_frame_rec {
    u32 err, nblocks, has_checksum, skippable;
    u64 fcs;
    u32 end_pos;
    u32 cost;               // sequences in all blocks: the sequence lanes' scheduling key (zl2_order_kernel)
    u32 lit_cost;           // Huffman-coded literal bytes in all blocks: the literal lanes' scheduling key
    u32 pad2;
};

// general kernel: resident workgroups per CU = scratch slots per CU (80 VGPRs; at 4 and 128 VGPRs dictionary frames of 64 KiB
// took 25.9 instead of 19.6 ms per 4000 and 100 000 small ones 4.9 instead of 4.1; 7 and 8 spill too much: 8.3)
#define ZL_DEC_WGS_PER_CU 6
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
    const zl2_frame_plan* plan;  // optional: with frec, only frames routed to this kernel are decoded
    const zl2_frame_rec* frec;
    u64* stats;  // optional (profiling): per-phase shader cycles summed over blocks, see zl_decode.hip
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
    u32 sec[5];       // where the entropy section's pieces start inside the dictionary buffer: Huffman description, offset-code,
                      // match-length and literal-length table descriptions, the three repeat offsets (zstd.c:42053-42150)
    u32 pad;
};

// A dictionary's tables as the lanes of zl_decode2.hip keep them in LDS (LitGroupT::tab, SeqLane::ent), built once per context by
// zl2_dict_prebuild_kernel from the synthetic block: the inheritance rounds of a batch of SMALL frames then copy 9 KiB per frame
// instead of building four tables per group of eight frames.
struct zl2_dict_pre {
    u32 huf_ok, huf_log;      // huf_ok: 1 usable, 0 the description is corrupt, 2 a 12-bit table (the general kernel's)
    u32 seq_ok, seq_log[3];   // LL, OF, ML
    u32 pad[2];
    u16 tab[2048];
    u32 ent[1280];
};

struct zl2_args {
    const zl_frame_desc* frames;
    const zl2_frame_plan* plan;
    zl2_frame_rec* frec;
    zl2_blk_rec* brec;
    u32 nframes;
    const u8* src;
    u8* dst;
    u64* results;
    u8* lit_scratch;
    u32* seq_scratch;
    u32 ignore_checksum;
    u64* stats;  // optional (profiling): exec-kernel phase cycles {scan, literals, matches, flush, batches, blocks}
    u32* max_blocks;  // device words, zeroed before the launch: [0] the largest block count of a frame on the pipeline (set by zl2_index_kernel), [1] frames left to the general kernel, [2] the schedule (ZL2_MODE_*, set by zl2_order_kernel), [3] frames zl2_small_kernel finished
    u32* order;       // 2 x nframes entries: the frames sorted by decreasing sequence count, then by decreasing literal bytes (zl2_order_kernel)
    u32 fuse1;        // launches whose frames all have one block: bit 0 literal and sequence lanes in ONE kernel (zl2_entropy1_kernel); bit 1 a chain on a quad of lanes; bit 2 ... and on two waves, producer and consumer (round 5); 0: two kernels (experiments)
    // ---- large-frame pipeline only (zl_decode3.hip): the "frames" are segments (1 to ZL2_MAXB consecutive blocks) of ONE frame
    const zl2_blk_rec* ext;   // per segment 4 records of earlier blocks whose tables the segment inherits: [0] Huffman tree, [1..3] sequence tables in block order; type 0xFFFFFFFF: none.  Null on the ordinary path
    const struct zl3_big* big;  // execution state shared by the segments.  Null on the ordinary path
    // ---- frames that start from a DICTIONARY (zl_api.hip, batches on the ordinary path): its entropy tables are inherited the
    // way a segment inherits an earlier block's -- ext then holds ONE set of 4 records for every frame (ext_shared), describing a
    // synthetic block at ext_base: the dictionary's Huffman description, and a sequences header that spells out its three FSE
    // tables --, its content is history in front of position 0 of every frame, its repeat offsets are the history a frame starts with
    u32 ext_shared;
    const u8* ext_base;       // where the ext records' offsets count from (null: the frame's / segment's own source)
    const u8* dict_end;       // one past the dictionary's content (null: none)
    u32 dict_len;
    u32 dict_id;              // what a frame's Dictionary_ID must be when it has one (0: the context has no dictionary)
    u32 dict_rep[3];          // the history a frame starts with (1, 4, 8 without a formatted dictionary)
    const struct zl2_dict_pre* dict_pre;  // the dictionary's tables in the lanes' own formats, built once per context (null: the inheritance rounds build them from the synthetic block)
    const u8* dict_ext_base;  // large frames (zl_decode3.hip): where the offsets of an ext record that describes the dictionary count from (zl_decode2.hip ext_src), or null
    // ---- a launch may cover a PART of the cost-sorted order (zl2_launch deals a large batch out in parts, see there).
    // 0: all of it.  Otherwise the launch's ROLE in the two-part schedule; which groups a role covers depends on the mode
    // the order kernel chose on the device (max_blocks[2]: ZL2_MODE_*).
    u32 part;
    // large-frame pipeline: the first block of segment 0 starts from a history other than 1, 4, 8 (a formatted dictionary's, or
    // what the run in front of this run of blocks left behind): its chain writes symbols like every other block's, zl3_offsets_kernel
    // resolves them.  (Until round 3 that chain resolved its repeat offsets against 1, 4, 8 whatever the dictionary said;
    // dictionaries from ZDICT_trainFromBuffer say 1, 4, 8.)
    u32 sym_first;
};
#define ZL2_ROLE_E1 1u   // entropy kernels, main stream
#define ZL2_ROLE_ES 3u   // entropy kernels, second stream
#define ZL2_ROLE_X1 4u   // exec kernel, main stream (behind E1)
#define ZL2_ROLE_XS 5u   // exec kernel, second stream (behind ES)
#define ZL2_MODE_WHOLE 0u   // the batch is not cut: E1 and X1 cover everything, the other launches return at once
#define ZL2_MODE_NONE 2u    // every frame was decoded in front of the lanes (zl2_small_kernel): every launch returns at once
#define ZL2_MODE_HALVES 1u  // single-block frames: the expensive half of the order on the main stream (E1, X1), the cheap half -- entropy and exec -- beside it (ES, XS)
// the groups (of ZL2_WG_FRAMES frames of the sequence order) a launch covers: [g0, g1)
__device__ __forceinline__ void zl2_part_range(const zl2_args& a, u32& g0, u32& g1, bool& split) {
    const u32 all = (a.nframes + 7) / 8;
    g0 = 0; g1 = all; split = false;
    if (a.big == nullptr && a.max_blocks[2] == ZL2_MODE_NONE) { g1 = 0; return; }
    if (!a.part) return;
    const u32 mode = a.max_blocks[2];
    split = mode != ZL2_MODE_WHOLE;
    const bool main_role = a.part == ZL2_ROLE_E1 || a.part == ZL2_ROLE_X1;
    if (mode == ZL2_MODE_HALVES) { if (main_role) g1 = all / 2; else g0 = all / 2; }
    else if (!main_role) g1 = 0;
}
// Large-frame pipeline: what the segment workgroups of zl2_exec_kernel<true> share.
#define ZL3_ROOT 0xFFFFFFFFu      // link value of a byte that is final in the output (marks of later rounds: ZL3_ROOT - 1, - 2)
#ifndef ZL3_EXEC_RUN
#define ZL3_EXEC_RUN 8            // consecutive segments one exec workgroup works through (zl3_big::exec_run)
#endif
#define ZL3_TILE_LOG 14           // resolve rounds skip 16 KiB tiles without pending bytes
struct zl3_info {                 // one per large frame, device memory
    u32 err;                      // first error of any stage (0: none)
    u32 nblocks, nsegs;
    u32 has_checksum;
    u64 fcs;
    u32 end_pos;                  // where the checksum sits
    u32 hdr_size;                 // frame header bytes (chunk-hinted walk)
    u64 lit_total, seq_total;     // scratch the segments need: literal bytes, 16-byte sequence records
    u64 produced;                 // regenerated bytes (zl3_offsets_kernel)
    u32 pending;                  // (unused)
    u32 ndependent;               // blocks that ride on the tables of an earlier block (treeless literals, Repeat_Mode)
    u32 rep_end[3];               // the repcode history behind the last block (zl3_offsets_kernel): what a following run of the same frame starts with
    // a run of a frame's blocks (zl3_hint::ndonors): the blocks in front of the run whose table descriptions are in force at its
    // start sit behind the run's own in the block table -- parsed like them, decoded never -- and donor[] says which of them
    // (index into the table, or 0xFFFFFFFF) holds the Huffman tree, the LL, OF and ML table
    u32 ndonors;
    u32 donor[4];
};
struct zl3_big {
    zl3_info* info;
    u32* link;                    // per output byte: ZL3_ROOT (or a later round's mark) when final, else the frame position it copies
    u32* tile_flag;               // per 16 KiB of output: some byte is pending
    const u32* blk_dst;           // per block: first output position (nblocks + 1 entries)
    const u32* rep_start;         // per block: repcode history it starts with (3 words)
    const u32* seg_first;         // per segment: its first block
    u32 exec_run;                 // consecutive segments one exec workgroup works through
    u32 dict_len;                 // bytes of dictionary content that sit, as history, in front of the frame's first byte (0: none)
    const u8* dict_end;           // one past the last of them: frame position -k is dict_end[-k]
};

struct zl2_side {           // owned by the decompression context: lets the two entropy kernels run side by side
    hipStream_t stream;
    hipEvent_t fork, join;
    hipStream_t part_stream[3];  // the cheaper parts of a large batch run here, entropy and exec, next to the expensive part's chains
    hipEvent_t part_join[3];
};
extern "C" hipError_t zl2_launch(const zl2_args* a, hipStream_t stream, const zl2_side* side, hipEvent_t* ev);
// large-frame pipeline: the entropy kernels / the exec kernel over the segments of one frame
extern "C" hipError_t zl2_launch_segments_entropy(const zl2_args* a, hipStream_t stream, const zl2_side* side);
extern "C" hipError_t zl2_launch_segments_exec(const zl2_args* a, u32 exec_run, hipStream_t stream);

extern "C" hipError_t zl_launch_decode(const zl_decode_args* a, u32 nslots, hipStream_t stream);
extern "C" hipError_t zl2_launch_dict_prebuild(const zl2_blk_rec* ext, const u8* ext_base, struct zl2_dict_pre* out, hipStream_t stream);
extern "C" hipError_t zl_launch_digest_dict(const u8* d_dict, u32 n, void* d_tables, hipStream_t stream);
