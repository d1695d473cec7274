/*
 * zl_hip.h -- C-ABI of libzl_hip.so: the MI355X (gfx950) zstd one-shot compress/decompress
 * path that sits where the reference package's R/C glue calls its bundled libzstd.
 *
 * Plain C, plain pointers and sizes; no torch/R types.  Every entry point below names the
 * reference interface it replaces (paths relative to the reference repository; "zstd.h" is
 * src/zstd/zstd.h, "zstd.c" is src/zstd/zstd.c).  INTEGRATION.md shows the two-line change
 * in src/Makevars + src/raw-file.c / src/serialize.c / src/cctx.c / src/dctx.c that binds
 * them.
 *
 * Conventions kept from libzstd so that the glue's error handling works unchanged
 * (src/raw-file.c:81-83, :190-192):
 *   - size_t results; an error is (size_t)-code with code from ZSTD_ErrorCode
 *     (zstd.c:1463-1499), detected by zl_isError (zstd.c:1548), described by
 *     zl_getErrorName with the same strings (zstd.c:3585-3628) -- the string
 *     "Restored data doesn't match checksum" is asserted by the reference's
 *     tests/testthat/test-checksums.R:24-27.
 *   - the caller owns src and dst; nothing is retained after a call returns; no call
 *     leaves device work in flight (the glue may longjmp via Rf_error right after).
 *   - one caller thread per context.
 *
 * There is NO CPU fallback: every compress/decompress entry point runs HIP kernels and
 * returns an error code (ZL_error_GPU_unavailable == memory_allocation, 64) when no gfx950
 * device can be initialised.
 */
#ifndef ZL_HIP_H
#define ZL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zl_cctx_s zl_cctx; /* replaces ZSTD_CCtx (zstd.h:266) */
typedef struct zl_dctx_s zl_dctx; /* replaces ZSTD_DCtx (zstd.h:278) */

/* parameter ids: numerically identical to ZSTD_cParameter / ZSTD_dParameter so that
 * src/cctx.c:265-284 and src/dctx.c:99,163 compile unchanged against the twins */
#define ZL_c_compressionLevel 100 /* zstd.h:333  ZSTD_c_compressionLevel.  Operating points: negative levels and 1, 2 (every other position
                                   * probed; negative levels store their literals and skip eagerly), 3 (the double-hash parse; 0 means 3), 4
                                   * (larger match tables, longer verification), 5 (... denser far search, two-step lazy rule, all three repeat
                                   * offsets), 6 ... 15 (the deep match finder: the reference's own search depth -- 8 / 16 / 16 / 16 / 32 / 64 /
                                   * 96 / 128 / 192 / 256 candidates per position --, lazy depth -- one step at 6 and 7, two from 8 -- and
                                   * window -- 2 MiB up to 8, 4 MiB from 9 -- at that level, zstd.c:29523-29550).  16 ... 22: level 15's search inside the reference's window at
                                   * that level -- 4 / 8 / 8 / 8 / 32 / 64 / 64 MiB (the reference's 128 MiB at 22 is cut to a section) -- and, round 6, the
                                   * PRICE-BASED parse (csrc/zl_opt.hip; the reference's btopt / btultra / btultra2, zstd.c:34586-36163): every position's
                                   * nearest offset per length, a forward pass over arrival prices taken from the chunk's own symbol statistics, twice.
                                   * ONE input of 16 KiB ... 16 MiB is searched by the deep finder at levels 3, 4
                                   * and 5 too (3 / 4 / 6 candidates, one step lazy: smaller output AND a shorter call than the tables give there;
                                   * the batch entries and dictionary contexts keep the tables).  zl_CCtx_getParameter reports the point a context runs at.  Contexts with a
                                   * dictionary and the batch entries run levels >= 6 on the table-based parse at its deepest setting (the deep
                                   * finder sorts the positions of ONE contiguous input). */
#define ZL_c_checksumFlag 201     /* zstd.h:440  ZSTD_c_checksumFlag     */
#define ZL_c_nbWorkers 400        /* zstd.h:449  ZSTD_c_nbWorkers (accepted, ignored: the GPU path is always parallel) */
#define ZL_c_stableInBuffer 1006  /* zstd.h:2083 ZSTD_c_stableInBuffer  (accepted, no-op) */
#define ZL_c_stableOutBuffer 1007 /* zstd.h:2103 ZSTD_c_stableOutBuffer (accepted, no-op) */
/* Frame layout of inputs larger than one 128 KiB chunk (new; no libzstd twin).
 *   ZL_layout_one_frame (2): ONE zstd frame, like ZSTD_compress2 writes (src/raw-file.c:74): the chunks become
 *     consecutive runs of blocks of that frame, matches reach up to 2 MiB back across chunks (window log 21 as the
 *     reference's level 3, zstd.c:29529), zl_findFrameCompressedSize(out) == bytes written, and a reader that takes
 *     the first frame only (src/raw-file.c:150-155, src/raw-file-in.c:66) gets everything.
 *   ZL_layout_frames (1): one independent frame per chunk, concatenated: decodes fastest (no chunk depends on
 *     another), but a first-frame reader sees 128 KiB, and nothing further back than the chunk is matched.
 *   ZL_layout_auto (0, default): one frame for zl_compress2 / zl_compressStream2, independent frames for the
 *     device-resident and batch entries.  Contexts with ZL_c_checksumFlag or a dictionary follow the same rule (round 3):
 *     one frame carries ONE content checksum over all of it (zstd.c:27733; XXH64 is a single dependency chain, so it is
 *     computed on a host thread beside the upload and checked on the host beside the download), and a dictionary's
 *     content is history in front of the frame's first chunk (zstd.c:27550). */
#define ZL_c_frameLayout 9003
#define ZL_layout_auto 0
#define ZL_layout_frames 1
#define ZL_layout_one_frame 2
/* Host entries (zl_compress2, zl_decompressDCtx) overlap the upload, the kernels and the download of large inputs
 * (sections of 64 MiB on several streams; the download runs on a second host thread).  1 (default) / 0: one after the
 * other, as round 1 did -- kept for measurements.  The bytes written are the same either way. */
/* zl_compress2 appends a seek table: a skippable frame (zstd.h:138; ignored by every standard decoder,
 * zstd.c:41722-41733, and by a reader that takes the first frame only) listing the compressed size of every 128 KiB
 * chunk.  zl_decompressDCtx and zl_index_frames find it from the end of the buffer and skip the serial walk over
 * every block header of the data in front of it; the table is checked against the data, never trusted.  Off by default:
 * the output is then exactly one frame (one-frame layout) and zl_findFrameCompressedSize(out) == bytes written -- except
 * for ONE frame of 4 GiB of content or more, which always carries the table: this library's large-frame decoder plans frames
 * below 4 GiB (32-bit positions); a larger one is decoded section by section from the table.  (Without it -- the device
 * entries' chunk hint, or a frame another writer made -- such a frame is decoded in runs of blocks: ZL_d_runBlocks below.)
 * Independent frames (ZL_layout_frames): the table is zstd's published SEEKABLE FORMAT (contrib/seekable_format of the zstd
 * distribution: skippable magic 0x184D2A5E, {compressed size, decompressed size} per frame, footer magic 0x8F92EAB1), which
 * that format's other readers use as well.  ONE frame: the published format lists frames, not the runs of blocks inside
 * one, so the table there is private to this library (zl_format.h: trailer tag "ZLIX"): other tools skip it. */
#define ZL_c_seekTable 9006
/* Devices behind one call: zl_compress2 / zl_decompressDCtx of a large host buffer deal its 64 MiB sections out
 * over this many device slots -- slot s on device s mod hipGetDeviceCount() --, each with its own host thread, upload and
 * download (its own PCIe link; nothing is funnelled through one device).  1 (default): the current device only.  0: every
 * visible device, one per 512 MiB of input at least.  The bytes written do not depend on the number of devices.
 * Decompression spreads independent frames, and one frame of this library when a seek table says where its chunks are. */
#define ZL_c_devices 9007
#define ZL_d_devices 9008
#define ZL_c_pipeline 9004
#define ZL_d_pipeline 9005
/* Frames of 4 GiB and more that carry no chunk hint or seek table (what the reference writes for a large R object) are decoded
 * in runs of at most 8192 blocks, each run with the output in front of it as its history (zl_api.hip
 * decode_walked_frame_in_runs).  ZL_d_runBlocks = N (16 .. 8192; 0 = default) makes every frame of more than N blocks that a
 * host entry decodes take that path, in runs of at most N blocks: how tests and fuzzers reach it with frames of a few MiB. */
#define ZL_d_runBlocks 9009
/* ONE frame of 256 MiB of content or more that zl_decompressDCtx is handed in host memory moves slab by slab (round 4): a thread
 * uploads it piece by piece from the start of the call, it is decoded in runs of blocks as those arrive (the same runs, with the
 * output so far as history) and every run's bytes come down -- and are hashed, when the frame carries a checksum -- while the
 * next run is decoded.  zl_decompressStream does the same with a frame of 32 MiB or more while it ARRIVES.
 * ZL_d_slabMiB = N (0 = defaults: 256, and 32 in a stream) sets both thresholds in MiB: how tests reach the paths with small
 * frames (together with ZL_d_runBlocks for the size of the runs).  ZL_d_pipeline = 0 switches them off. */
#define ZL_d_slabMiB 9010
#define ZL_d_stableOutBuffer 1001     /* zstd.h:2422 ZSTD_d_stableOutBuffer (accepted, no-op) */
#define ZL_d_forceIgnoreChecksum 1002 /* zstd.h:2433 ZSTD_d_forceIgnoreChecksum */

#define ZL_CONTENTSIZE_UNKNOWN_ (0ULL - 1) /* zstd.h:190 */
#define ZL_CONTENTSIZE_ERROR_ (0ULL - 2)   /* zstd.h:191 */

/* ---- error helpers: ZSTD_isError / ZSTD_getErrorName (zstd.c:15758,15762) -------------- */
unsigned zl_isError(size_t code);
const char *zl_getErrorName(size_t code);

/* ---- sizes ------------------------------------------------------------------------------
 * zl_compressBound replaces ZSTD_compressBound (zstd.h:232; call sites src/raw-file.c:52,
 * src/serialize.c:73).  Covers every layout the encoder can write (one frame per 128 KiB chunk is the
 * largest: the sum of per-frame bounds) plus the optional seek table; always >= ZSTD_compressBound(srcSize). */
size_t zl_compressBound(size_t srcSize);

/* First-frame helpers, exact twins of ZSTD_findFrameCompressedSize (zstd.h:210) and
 * ZSTD_getFrameContentSize (zstd.h:192); call sites src/raw-file.c:150,155,
 * src/serialize.c:171,177. */
size_t zl_findFrameCompressedSize(const void *src, size_t srcSize);
unsigned long long zl_getFrameContentSize(const void *src, size_t srcSize);

/* Whole-buffer helpers for concatenated frames (what the independent-frames layout produces for inputs
 * larger than one chunk: device-resident and batch entries, or ZL_c_frameLayout = ZL_layout_frames).
 * zl_findDecompressedSize is the twin of
 * ZSTD_findDecompressedSize (zstd.h:1484): sum of the content sizes of all frames.
 * zl_findAllFramesCompressedSize: total bytes covered by complete frames from src[0]. */
unsigned long long zl_findDecompressedSize(const void *src, size_t srcSize);
size_t zl_findAllFramesCompressedSize(const void *src, size_t srcSize);

/* ---- compression context: ZSTD_createCCtx / ZSTD_freeCCtx / ZSTD_CCtx_setParameter /
 * ZSTD_CCtx_getParameter / ZSTD_CCtx_loadDictionary (zstd.c:22610, 22693, 23223, 23531,
 * 23826; call sites src/cctx.c:72-95,222,265-304,352-354) --------------------------------- */
zl_cctx *zl_createCCtx(void);
size_t zl_freeCCtx(zl_cctx *cctx);
size_t zl_CCtx_setParameter(zl_cctx *cctx, int param, int value);
size_t zl_CCtx_getParameter(const zl_cctx *cctx, int param, int *value);
size_t zl_CCtx_loadDictionary(zl_cctx *cctx, const void *dict, size_t dictSize); /* copied, as zstd.c:23826-23830 */

/* replaces ZSTD_compress2 (zstd.h:599; call sites src/raw-file.c:74, src/serialize.c:91):
 * host src -> host dst.  ONE complete zstd frame with the content size recorded, as the reference writes, whatever the
 * context holds -- level, checksum flag, dictionary (ZL_c_frameLayout selects independent frames instead).  Large
 * inputs are uploaded, compressed and downloaded in overlapping 64 MiB sections (ZL_c_pipeline), optionally
 * on several devices (ZL_c_devices); the bytes written depend on neither. */
size_t zl_compress2(zl_cctx *cctx, void *dst, size_t dstCapacity, const void *src, size_t srcSize);

/* ---- decompression context: ZSTD_createDCtx / ZSTD_freeDCtx / ZSTD_DCtx_setParameter /
 * ZSTD_DCtx_getParameter / ZSTD_DCtx_loadDictionary (zstd.c:40913, 40927, 42507, 42478,
 * 42321; call sites src/dctx.c:54,99,119,163,181-186,234) --------------------------------- */
zl_dctx *zl_createDCtx(void);
size_t zl_freeDCtx(zl_dctx *dctx);
size_t zl_DCtx_setParameter(zl_dctx *dctx, int param, int value);
size_t zl_DCtx_getParameter(const zl_dctx *dctx, int param, int *value);
size_t zl_DCtx_loadDictionary(zl_dctx *dctx, const void *dict, size_t dictSize);

/* FORMAT STRICTNESS (a stated divergence from libzstd on DAMAGED input): every decoder of this library requires each
 * entropy stream -- the four Huffman streams of a literals section, the sequence bitstream -- to be consumed exactly, as the
 * format document says.  libzstd's Huffman fast path (zstd.c:38890-38940) does not check that, so a few damaged frames that
 * ZSTD_decompressDCtx accepts (returning bytes that are not what was compressed) are REJECTED here with
 * "Data corruption detected" (ZL_E_CORRUPTION, libzstd's ZSTD_error_corruption_detected = 20).  Never the other way round: whatever this library accepts, libzstd accepts, with the same
 * bytes (tests/test_oracle.py, tests/test_decode_gpu.py).  On valid frames the results are identical. */

/* replaces ZSTD_decompressDCtx (zstd.h:290; call sites src/raw-file.c:189,
 * src/serialize.c:197): decodes every frame in [src, src+srcSize) -- concatenated small frames in
 * parallel on the lane pipeline, a frame of many blocks (what ZSTD_compress2 and zl_compress2 write for
 * large inputs) spread over the whole device by the large-frame pipeline, skippable frames skipped --
 * verifies content size and checksum, returns bytes regenerated. */
size_t zl_decompressDCtx(zl_dctx *dctx, void *dst, size_t dstCapacity, const void *src, size_t srcSize);

/* dictionary id helpers: ZSTD_getDictID_fromDict / ZSTD_getDictID_fromFrame
 * (src/dictionaries.c:23-58) */
unsigned zl_getDictID_fromDict(const void *dict, size_t dictSize);
unsigned zl_getDictID_fromFrame(const void *src, size_t srcSize);

/* ---- batch entries (new; SURVEY.md section 7 H1: the reference API is one buffer per
 * call, .Call()ed in an lapply loop -- vignettes/dictionaries.Rmd:129-138).  n independent
 * inputs, one launch.  results[i] = bytes written for item i or an error code.  Return
 * value: 0, or the first error met before any item ran (e.g. no GPU). ---------------------- */
size_t zl_compress_batch(zl_cctx *cctx, size_t n, void *const *dsts, const size_t *dstCapacities,
                         const void *const *srcs, const size_t *srcSizes, size_t *results);
size_t zl_decompress_batch(zl_dctx *dctx, size_t n, void *const *dsts, const size_t *dstCapacities,
                           const void *const *srcs, const size_t *srcSizes, size_t *results);

/* ---- device-resident entries (new): data already in HBM, used by bench.py and by callers
 * that keep buffers on the GPU.  Pointers are device addresses; `stream` is a hipStream_t
 * (NULL = the context's own stream).  The *_async calls only enqueue work; *_finish waits
 * for it and returns the total bytes or the first error.
 * NULL is NOT the legacy default stream: the context's own stream is non-blocking, i.e. not ordered behind work queued on
 * the default stream.  A caller that produces d_src on the default stream either passes a stream of its own that is, or
 * synchronises first (zstdlite_amd/device.py does the latter). ----------------------------- */
typedef struct {
    uint64_t src_off;  /* frame start inside d_src */
    uint64_t src_size; /* compressed size of the frame */
    uint64_t dst_off;  /* where its content goes inside d_dst */
    uint64_t dst_cap;  /* bytes available there */
} zl_frame_t;

/* Decode nframes frames described by the host table `frames` from d_src into d_dst.
 * Frames of more than eight blocks are decoded one after the other by the large-frame pipeline, and this call waits for
 * each of them (it reads their block counts back); for such a frame with a content checksum -- unless
 * ZL_d_forceIgnoreChecksum is set -- the regenerated bytes are also brought to the host in 16 MiB pieces and hashed there:
 * XXH64 is one serial chain, 0.5 GB/s on GPU lanes against 25 GB/s and more on a CPU core.  Frames of
 * 4 GiB and more are decoded in runs of blocks (ZL_d_runBlocks above); without zl_decompress_device_hint_host_copy their
 * COMPRESSED bytes are brought down once for the run plan. */
size_t zl_decompress_device_async(zl_dctx *dctx, void *d_dst, const void *d_src, const zl_frame_t *frames,
                                  size_t nframes, void *stream);
/* results (host, may be NULL) receives the per-frame regenerated sizes / error codes. */
size_t zl_decompress_device_finish(zl_dctx *dctx, size_t *results);
/* Optional, before zl_decompress_device_async: frames[0] of the next call is ONE frame written by
 * zl_compress_device_async in the one-frame layout, and chunkSizes (nchunks entries, as zl_compress_device_finish
 * reported them) say where each 128 KiB chunk's run of blocks starts.  The decoder then finds the frame's blocks with a
 * lane per chunk instead of following 3-byte headers through the whole frame on one lane (zstd.c:43075).  Only a hint:
 * it is checked against the frame and ignored when it does not fit; it is forgotten after one call. */
size_t zl_decompress_device_hint(zl_dctx *dctx, const size_t *chunkSizes, size_t nchunks);
/* Optional, before zl_decompress_device_async: the caller still holds the compressed bytes in host memory (hostSrc[0..n)
 * equals d_src[0..n); typically it read a file and uploaded it).  The block headers of large frames are then walked on
 * the host, as zl_findFrameCompressedSize does (zstd.c:41335), instead of by one GPU lane.  Forgotten after one call. */
size_t zl_decompress_device_hint_host_copy(zl_dctx *dctx, const void *hostSrc, size_t n);

/* Compress d_src[0..srcSize) into d_dst: concatenated frames, one per 128 KiB chunk (the default of this
 * entry), or ONE frame (ZL_c_frameLayout = ZL_layout_one_frame).  d_dst must hold zl_compressBound(srcSize)
 * bytes.  After _finish, frameSizes (host, may be NULL, capacity zl_compress_device_nframes(srcSize)) holds
 * the compressed size of every chunk -- its frame, or its run of blocks inside the one frame (the first
 * includes the frame header): what zl_decompress_device_hint takes.  Returns total compressed bytes.
 * One configuration is NOT asynchronous: ONE frame with ZL_c_checksumFlag.  The frame's content checksum is one XXH64 over
 * all of the input (zstd.c:27733), a serial chain that runs on a host core; zl_compress_device_async brings the input down
 * and hashes it before it returns, so that on return the frame in d_dst is complete and d_src may be reused -- exactly as
 * for every other configuration a stream-ordered consumer may assume. */
size_t zl_compress_device_nframes(size_t srcSize);
size_t zl_compress_device_async(zl_cctx *cctx, void *d_dst, size_t dstCapacity, const void *d_src, size_t srcSize,
                                void *stream);
size_t zl_compress_device_finish(zl_cctx *cctx, size_t *frameSizes);

/* Build a frame table for concatenated frames that live on the HOST (walks block headers as
 * ZSTD_findFrameSizeInfo does, zstd.c:41335): fills up to cap entries, returns the number
 * of data frames found or an error code.  dst offsets are assigned back to back. */
size_t zl_index_frames(const void *src, size_t srcSize, zl_frame_t *frames, size_t cap);

/* ---- streaming entries: twins of ZSTD_compressStream2 / ZSTD_decompressStream and the
 * session calls around them, as the reference's file / connection glue uses them
 * (src/raw-file-out.c:74-113, src/raw-file-in.c:87-102, src/serialize-file-out.c:84-219,
 * src/serialize-file-in.c:90, src/zstdfile.c:202-238,362,475-511; zstd.h:800-880, :930-960).
 * The buffer structs and the directive values are libzstd's, so a binding only renames.
 *
 * Compression: input is collected on the host into jobs of 64 MiB (libzstd's own multi-threaded
 * streaming cuts jobs the same way, zstd.c:36848); every full job -- and the rest at ZL_e_flush /
 * ZL_e_end -- goes through the same kernels as zl_compress2.  With the size pledged
 * (zl_CCtx_setPledgedSrcSize, what src/raw-file-out.c:74 does) the jobs are consecutive runs of blocks of
 * ONE frame: byte for byte what zl_compress2 writes for the whole input, so a reader that sizes its
 * buffer from the frame header (src/raw-file-in.c:66) or takes the first frame gets everything.
 * Inside such a session ZL_e_flush cuts no run: the frame's runs end at 64 MiB boundaries and at the
 * pledged end only, so a flush hands out the jobs that are complete and keeps collecting the rest
 * (the reference's glue never flushes; a frame of its own in mid-frame would be an invalid stream).
 * So in a pledged session NOT everything consumed so far can be decoded from what a flush hands out: up to one
 * job of input stays collected until its boundary or the end, and ZL_e_flush says so -- the first flush after
 * input has arrived returns the bytes that wait to be handed out PLUS the input bytes that stay collected
 * (non-zero: zstd.h:841-880 reads that as "not fully flushed").  The same flush repeated without new input
 * returns only the former, so that the loop zstd.h documents ("call again until it returns 0") ends.
 * Without a pledge every job is a frame of its own with its own recorded size: valid zstd that
 * ZSTD_decompressStream / ZSTD_decompress read frame after frame.
 * A full job runs on a host thread of the context's own while the caller's next pieces are collected
 * (one job at a time): its output is handed out by the call that fills the next job, or by ZL_e_flush /
 * ZL_e_end, whichever comes first.  Every other entry that takes the context waits for that job first;
 * the context is still one-caller-at-a-time.
 * Returns the bytes still waiting to be handed out (0 at ZL_e_end: the session is complete), or an
 * error code.  Decompression: input is collected until it holds complete frames; those are decoded
 * at once (all routes of zl_decompressDCtx, frames without a recorded size included) -- straight into
 * the caller's buffer when it has room for all of them -- and handed out as the output buffer allows.
 * A frame whose header records 32 MiB of content or more (ZL_d_slabMiB; no dictionary) is not
 * collected: every call uploads the bytes it is handed, the frame is decoded in runs of blocks as they
 * complete and the output comes down into the caller's buffer as the runs finish, so the first bytes are
 * out long before the last ones are in (round 4; until then a 1 GiB frame was buffered whole).  A call
 * that brings no input waits for what is in flight.  Returns 0 when everything received so far has been decoded and
 * handed out, an error code, or a non-zero hint.  Sixteen calls in a row without progress return
 * the no-forward-progress errors of zstd.c:42957-42960. */
typedef struct { const void *src; size_t size; size_t pos; } zl_inBuffer;  /* ZSTD_inBuffer  (zstd.h:805) */
typedef struct { void *dst; size_t size; size_t pos; } zl_outBuffer;       /* ZSTD_outBuffer (zstd.h:811) */
typedef enum { ZL_e_continue = 0, ZL_e_flush = 1, ZL_e_end = 2 } zl_EndDirective;                 /* ZSTD_EndDirective (zstd.h:841) */
typedef enum { ZL_reset_session_only = 1, ZL_reset_parameters = 2, ZL_reset_session_and_parameters = 3 } zl_ResetDirective; /* zstd.h:575 */
size_t zl_compressStream2(zl_cctx *cctx, zl_outBuffer *output, zl_inBuffer *input, zl_EndDirective endOp);
size_t zl_CCtx_setPledgedSrcSize(zl_cctx *cctx, unsigned long long pledgedSrcSize); /* checked at ZL_e_end (zstd.c:23793) */
size_t zl_CCtx_reset(zl_cctx *cctx, zl_ResetDirective reset);
size_t zl_CStreamInSize(void);  /* ZSTD_CStreamInSize: 128 KiB; any size works, jobs are cut inside */
size_t zl_CStreamOutSize(void);
size_t zl_decompressStream(zl_dctx *dctx, zl_outBuffer *output, zl_inBuffer *input);
size_t zl_DCtx_reset(zl_dctx *dctx, zl_ResetDirective reset);
size_t zl_DStreamInSize(void);
size_t zl_DStreamOutSize(void);

/* ---- diagnostics (new): per-kernel device time of the last device-resident call, measured with
 * HIP events recorded on the stream the kernels were launched on.  Enable with
 * zl_CCtx_setParameter(cctx, ZL_c_profile, 1) / zl_DCtx_setParameter(dctx, ZL_d_profile, 1).
 * zl_cctx_kernel_ms fills ms[0..3] = {match, entropy, size-scan, gather}; zl_dctx_kernel_ms
 * fills ms[0..4] = {whole decode, frame/block index, literal+sequence lanes, execution, general
 * kernel}.  Return 0, or an error code when profiling was not enabled.  A value of 2 additionally makes
 * the kernels count shader cycles per phase (global atomics: slows them down; development only). */
#define ZL_c_faultInjection 9098 /* TESTS ONLY (the library reads no environment variables): bit 0 -- the deep match finder's allocations "fail", so that the
                                  * call is made again on the table parse (levels >= 6 degrade instead of failing); bits 4 and up: k -- from the k-th full job
                                  * of a streaming session on, the job's thread "cannot be had" and the job runs on the caller (tests/test_stream_gpu.py,
                                  * tests/test_compress_gpu.py).  0: off. */
#define ZL_c_profile 9001
#define ZL_d_profile 9002
size_t zl_cctx_kernel_ms(const zl_cctx *cctx, double *ms);
size_t zl_cctx_far_ms(const zl_cctx *cctx, double *ms); /* the far-table passes in front of the match kernel (one-frame layout; 0 otherwise) */
size_t zl_dctx_kernel_ms(const zl_dctx *dctx, double *ms);
/* Frames this context has decoded with the large-frame pipeline (one frame of many blocks spread over
 * the whole device: what ZSTD_compress2 writes for inputs above 128 KiB, src/raw-file.c:74) rather
 * than with one workgroup.  Lets tests and benchmarks assert which path ran. */
size_t zl_dctx_large_frames(const zl_dctx *dctx);
/* ... and the runs of blocks that frames of 4 GiB and more (or, with ZL_d_runBlocks, of more than N blocks) were decoded in. */
size_t zl_dctx_frame_runs(const zl_dctx *dctx);
/* Device slots the last zl_compress2 / zl_decompressDCtx call spread its work over (ZL_c_devices / ZL_d_devices): 1 when it ran on
 * the current device alone.  Lets tests assert that the sharded path ran. */
size_t zl_cctx_last_slots(const zl_cctx *cctx);
size_t zl_dctx_last_slots(const zl_dctx *dctx);

/* library / device info */
const char *zl_versionString(void); /* ZSTD_versionString twin for src/serialize.c:26 */
int zl_deviceCount(void);

#ifdef __cplusplus
}
#endif
#endif /* ZL_HIP_H */
