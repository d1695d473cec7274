// zl_encode.h -- launch-argument structs for the encoder kernels (zl_encode.hip) shared with the
// C-ABI host code (zl_api_compress.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "zl_format.h"

#define ZL_CHUNK ZL_BLOCK_MAX                    // one frame per 128 KiB of input
#define ZL_MATCH_TABLE_BYTES (((1 << 13) + (1 << 12)) * 2)  // 16-bit entries of the long and the short table
#define ZL_CDICT_BYTES 8192                      // device-side encoder tables of a dictionary (struct zl_cdict, zl_encode.hip)
#define ZL_DICT_REACH 65000                       // trailing dictionary bytes the 16-bit entries can address
#define ZL_ENC_BLOCKS 8                          // blocks per frame (they share one set of entropy tables)
#define ZL_SEQ_STRIDE ((ZL_CHUNK / 4) + 8)       // sequences per chunk (min match 4)
#define ZL_LIT_STRIDE_ENC (ZL_CHUNK + 64)
#define ZL_ENT_HANDOVER 12288                   // bytes of shared state the two halves of the entropy stage hand over per chunk (zl_encode.hip EntShared behind its scratch regions)
#define ZL_SST_BYTES 9                          // per sequence slot of the FSE-chain scratch: 3 code bytes + 3 16-bit states
#define ZL_FRAME_OVERHEAD 24                     // magic 4 + FHD 1 + dictID 4 + FCS 4 + block header 3 + checksum 4 (+ slack)
#define ZL_OUT_STRIDE (ZL_CHUNK + ZL_FRAME_OVERHEAD + 8)

// ---- far matches (single-frame "windowed" layout): a hash table in HBM over SAMPLED positions of the whole input,
// built before the match kernel runs, gives every 32-byte neighbourhood at most one offset worth trying
// that reaches further back than the chunk's own LDS tables (up to ZL_FAR_GROUP - 1 chunks = the frame's window).
#ifndef ZL_FAR_SAMPLE_LOG
#define ZL_FAR_SAMPLE_LOG 3                      // one position in 8 is inserted / probed (content-defined: top hash bits zero)
#endif
#ifndef ZL_FAR_BUCKET_LOG
#define ZL_FAR_BUCKET_LOG 14
#endif
// ZL_FAR_BUCKET_LOG:                     // buckets per group of chunks
#define ZL_FAR_GROUP 16                          // a chunk looks into the ZL_FAR_GROUP - 1 chunks in front of it
#define ZL_FAR_WLOG 21                           // window the frame header declares: ZL_FAR_GROUP chunks of 128 KiB (the reference's level 3 uses the same, zstd.c:29529)
#ifndef ZL_FAR_HINT_LOG
#define ZL_FAR_HINT_LOG 4                        // one hint word per 16 input bytes
#endif
#define ZL_FAR_SECTION_LOG 9                      // far history does not cross sections of 512 chunks (64 MiB): a section is what the host entries
                                                 // upload, compress and download as one pipeline stage, and the output must not depend on how a call was staged
#define ZL_FAR_TABLE_BYTES(nchunks) ((size_t)(nchunks) * ((size_t)4 << ZL_FAR_BUCKET_LOG))   // one column of 2^ZL_FAR_BUCKET_LOG words per chunk
#define ZL_FAR_LIST_CAP 40960u                    // sampled positions a chunk may list (expected: 16384 at one in eight, 32768 at one in four); what exceeds it is not probed
#define ZL_FAR_SPARSE_CAP 255u                    // ... and the keys of one listed position in 256 (by tag), which zl_far_probe_kernel tries first
#define ZL_FAR_LIST_WORDS (ZL_FAR_LIST_CAP * 2u + 16u + ZL_FAR_SPARSE_CAP + 1u)  // per chunk: 16 words of header, the pairs, the sparse keys' count and the keys
#define ZL_FAR_LIST_BYTES(nchunks) ((size_t)(nchunks) * (size_t)ZL_FAR_LIST_WORDS * 4)
#define ZL_FAR_HINT_WORDS (ZL_CHUNK >> ZL_FAR_HINT_LOG)

#define ZL_TRACE_BYTES (1 << 20)                 // development builds (-DZL_TRACE=<chunk>): timeline of that chunk's match workgroup
#define ZL_TRACE_POINTS 16                       // stamps per wave and iteration
#define ZL_TRACE_ITERS ((ZL_TRACE_BYTES / 8) / (4 * ZL_TRACE_POINTS))

struct zl_chunk_desc {
    u64 src_off;   // first byte of the chunk inside the source buffer
    u32 src_size;  // <= ZL_CHUNK
    u32 flags;     // bit 0: the chunk is a whole input of its own (at most one chunk): look for repeats inside the match window too
};

struct zl_seq {
    u32 lit_cum;    // literals emitted before this sequence (cumulative within the chunk)
    u32 match_len;
    u32 offset;
    u32 pos;        // position of the match inside the chunk
};

struct zl_chunk_meta {
    u32 nseq;
    u32 nlit;
};

struct zl_match_args {
    const zl_chunk_desc* chunks;
    const u8* src;
    u8* lits;         // per chunk lit_stride bytes
    zl_seq* seqs;     // per chunk seq_stride records
    zl_chunk_meta* meta;
    u32 lit_stride;
    u32 seq_stride;
    u64* stats;  // optional (profiling): cycles {find, parse, emit, insert, iterations}
    u32 sparse_only;  // levels below 3: the stride-2 body throughout (faster, lower ratio on match-dense data)
    u32 hi;           // levels >= 5: larger match tables, longer per-lane verification (slower, higher ratio)
    u32 lazy2;        // levels >= 5: the lazy rule also looks two positions on (zstd.c:31546 ZSTD_compressBlock_lazy_generic, depth 2)
    u32 dict_len;     // dictionary content bytes that sit, as history, in front of every chunk (0: none)
    const u8* dict;   // those bytes
    const u32* dict_hash;  // the two match tables as they look after the dictionary's last byte (ZL_MATCH_TABLE_BYTES)
    u32 intra;             // some chunk of the launch has flags bit 0 set (selects the kernel instantiation that can honour it)
    u32 accel;             // negative levels (-accel): windows without a match are skipped (1 + accel) times as eagerly (cf. the step of ZSTD_fast, zstd.c:29340: targetLength = -level)
    const u32* far_hints;  // windowed layout only: per chunk ZL_FAR_HINT_WORDS words, [25:0] an offset worth trying around these 32 bytes (0: none)
    // levels >= 6 (zl_deep.hip): the best earlier occurrence of every position, found before this kernel runs
    const u64* deep_best;  // one word per input byte from position deep_lo on (ZL_DEEP_* below), or null
    u64 deep_lo;           // input position of deep_best[0]
    u64 total;             // bytes of the whole input (the last readable byte of a chunk's section lies below it)
    u32 joined;            // the chunks are runs of blocks of ONE frame: a source may lie in front of its chunk (inside the chunk's section)
    u32 sub;               // deep parse of a launch that would not fill the device: 2, 4 or 8 workgroups share a chunk (zl_match_deep_kernel), 0 / 1: one each
    const u32* order;      // launch order (zl_launch_chunk_order): workgroup w takes chunk order[zl_spread(w)], or null: chunk zl_spread(w)
    // sub > 1: where the parts of a chunk leave their literals and sequences (laid out as lits / seqs) and their counts (2 words per part), for zl_deep_join_kernel
    u8* sub_lits;
    zl_seq* sub_seqs;
    u32* sub_meta;
};

// ---- the deep match finder (levels >= 6; zl_deep.hip).  Restates what the reference's lazy strategies search -- the most recent
// positions with the same hash of minMatch bytes, up to 2^searchLog of them, inside the window (zstd.c:31546-33746: hash chains
// and the row hash) -- for a machine that cannot walk the input in order: the positions of a section are SORTED by that hash
// (stable: ascending positions inside a hash value), so a position's K predecessors in the sorted order ARE its K most recent
// earlier occurrences; a workgroup takes 256 consecutive sorted entries, fetches 64 bytes around every position once, and all
// comparisons run out of LDS.  Result: one word per position -- the best candidate by the lazy parsers' gain 4 * length - log2(offset).
#define ZL_DEEP_HLOG 24        // bits of the hash the sort looks at: three passes of the radix sort (all 32 bits -- equal hashes are then equal minMatch bytes, and a position without a twin in the sorted order is never fetched -- cost a fourth pass, 0.35 ms per section, and spared the fetches of random bytes only: 1 GiB mix 73.3 -> 74.0 ms, text 17.6 -> 19.4 ms per 256 MiB)
#define ZL_DEEP_CAP 56         // bytes a candidate is measured over: a position's record holds [p - 8, p + 56)
#define ZL_DEEP_BACK 8         // ... and up to this many bytes in front of both
#define ZL_DEEP_EXT 1024       // a match that fills the record is measured on, this many bytes at most (zl_deep_extend_kernel)
// best word (64 bits): [15:0] length (0: no candidate), [23:16] bytes that agree in front of both, [63:32] offset
// Levels 16 ... 22 (zl_opt.hip) also want a position's CHAIN: among its K most recent twins, nearest first, every one that is longer
// than all nearer ones -- "the nearest offset per length", what the reference's binary tree hands its optimal parser (zstd.c:34300
// ZSTD_insertBtAndGetAllMatches).  The longest ZL_DEEP_CHAIN of them, ascending, the longest in the last slot; a word: [15:0] length, [63:32] offset, 0: none.
#define ZL_DEEP_CHAIN 4
struct zl_deep_args {
    const u8* src;     // the input (positions below are offsets into it)
    u64 lo, hi;        // the positions this call covers: whole chunks of ONE input, in order; sections of 2^(17 + ZL_FAR_SECTION_LOG) bytes are searched one after the other
    u64 total;         // bytes of the whole input
    u32 k;             // candidates per position
    u32 window;        // farthest offset
    u32 mml;           // bytes the hash covers (4, 5 or 6) = shortest match
    u32 frames;        // 1: independent frames -- a candidate must lie inside the position's own chunk
    void* scratch;     // zl_deep_scratch_bytes(hi - lo)
    size_t scratch_bytes;
    u64* best;         // out: one word per position, best[p - lo]
    u64* chain;        // out, or null: ZL_DEEP_CHAIN words per position, chain[ZL_DEEP_CHAIN * (p - lo) ...]
    u32 k2, mml2;      // with a chain: a second pass over every section, sorted by a hash of mml2 bytes, k2 candidates deep, merged into the chains (0: none)
};

struct zl_far_args {
    const zl_chunk_desc* chunks;  // the chunks of ONE contiguous input, in order
    const u8* src;
    u64 total;             // bytes of that input (positions whose 8 bytes would run past it -- or past their section -- are not hashed)
    u32 chunk_base;        // index, inside the input, of chunks[0] (a multiple of ZL_FAR_GROUP)
    u32 sample_log;        // one position in 2^sample_log takes part (ZL_FAR_SAMPLE_LOG at level 3; higher levels sample more densely)
    u32* table;            // ZL_FAR_TABLE_BYTES(nchunks): per chunk a column of 2^ZL_FAR_BUCKET_LOG words, written whole by zl_launch_far_build
    u32* lists;            // ZL_FAR_LIST_BYTES(nchunks): per chunk ZL_FAR_LIST_WORDS: a count (16 words), up to ZL_FAR_LIST_CAP pairs {bucket | tag << 14, position}, the sparse keys
    u32* hints;            // per chunk ZL_FAR_HINT_WORDS words, written by zl_launch_far_probe
};

struct zl_entropy_args {
    const zl_chunk_desc* chunks;
    const u8* src;
    const u8* lits;
    const zl_seq* seqs;
    const zl_chunk_meta* meta;
    u8* seq_states;       // scratch, per chunk seq_stride * ZL_SST_BYTES bytes: per table the sequences' codes (bytes) and FSE states (16-bit), see zl_entropy_kernel
    u8* out;           // per chunk out_stride bytes: the finished frame
    u32* frame_size;   // per chunk: bytes of the frame
    u32 lit_stride;
    u32 seq_stride;
    u32 out_stride;
    u32 checksum;      // append XXH64 low 32 bits (ZSTD_c_checksumFlag)
    u32 dict_id;       // 0 = none
    u32 raw_lits;      // negative levels: literals are stored, not Huffman-coded (zstd.c:20675 ZSTD_literalsCompressionIsDisabled: strategy fast with a target length, i.e. every negative level)
    u32 exact_rep;     // levels >= 5: the three-entry repeat-offset history of the decoder is followed sequence by sequence, codes 2 and 3 are emitted
    // single-frame layout: the chunks of this launch are consecutive runs of blocks of ONE frame
    u32 joined;        // 1: chunk 0 of the frame carries the frame header, only the frame's last chunk sets Last_Block, no per-chunk checksum
    u32 chunk_base;    // index, inside the frame, of this launch's first chunk
    u32 total_chunks;  // chunks of the whole frame
    u32 wlog;          // window the header declares when the content is larger than it
    u64 total_size;    // content size of the whole frame
    const void* cdict; // zl_cdict* (device) or null: the dictionary's entropy tables, offered to frames of a single block
    const u32* order;  // launch order of the chunks, as the match kernel's (or null)
    u8* handover;      // per chunk ZL_ENT_HANDOVER bytes: the stage runs as two kernels that hand their shared state over here (null: one kernel)
    u32 small_done;    // 1: zl_entropy_small_kernel runs first (small objects on a dictionary's tables, a wave each); zl_entropy_kernel does what it left (frame_size 0xFFFFFFFF)
    u64* stats;        // optional (profiling): cycles {statistics, tree | tables, codes + stream bits, block literals, block sequences, chains | tree description + tail, chunks}
};

struct zl_gather_args {
    const u8* staged;      // frames at out_stride
    const u32* frame_size;
    const u64* frame_off;  // exclusive scan of frame_size
    u8* dst;
    u32 out_stride;
};

extern "C" hipError_t zl_launch_match(const zl_match_args* a, u32 nchunks, hipStream_t stream);
extern "C" size_t zl_deep_scratch_bytes(u64 span);  // for a search over `span` bytes of input
extern "C" hipError_t zl_launch_deep_search(const zl_deep_args* a, hipStream_t stream);
extern "C" hipError_t zl_launch_match_deep(const zl_match_args* a, u32 nchunks, hipStream_t stream);
// levels 16 ... 22: the price-based parse over the chains, `passes` times (zl_opt.hip); the lazy parse has run (its sequences seed the statistics);
// arrivals: scratch, one word per input byte from deep_lo on (the best words' buffer: the lazy parse is done with it)
extern "C" hipError_t zl_launch_opt_parse(const zl_match_args* a, const u64* chain, u64* arrivals, u32 nchunks, u32 passes, hipStream_t stream);
// order[0 .. nchunks): the chunks by decreasing estimated cost (longest first), input order inside a class; scratch: nchunks words
extern "C" hipError_t zl_launch_chunk_order(const zl_chunk_desc* chunks, const u8* src, u32 nchunks, u32* cost_scratch, u32* order, hipStream_t stream);
extern "C" hipError_t zl_launch_far_build(const zl_far_args* a, u32 nchunks, hipStream_t stream);
extern "C" hipError_t zl_launch_far_probe(const zl_far_args* a, u32 nchunks, hipStream_t stream);
extern "C" hipError_t zl_launch_dict_hash(const u8* d_content, u32 n, u32* d_tables, hipStream_t stream);
extern "C" hipError_t zl_launch_cdict_build(const void* d_dict_tables, void* d_cdict, hipStream_t stream);
extern "C" hipError_t zl_launch_entropy(const zl_entropy_args* a, u32 nchunks, hipStream_t stream);
extern "C" hipError_t zl_launch_scan_sizes(const u32* frame_size, u64* frame_off, const u64* total_in, u64* total_out, u32 nchunks, hipStream_t stream);
extern "C" hipError_t zl_launch_gather(const zl_gather_args* a, u32 nchunks, hipStream_t stream);
