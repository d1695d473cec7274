// zl_api_compress.hip -- compression half of the C-ABI (include/zl_hip.h): the context the
// reference builds in init_cctx_with_opts (src/cctx.c:213-315) and the one-shot call it
// makes at src/raw-file.c:74 / src/serialize.c:91, on HIP kernels (zl_encode.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/zl_hip.h"
#include "zl_encode.h"
#include "zl_format.h"
#include "zl_host.h"
#include "zl_copy_pair.h"
#include "zl_kernels.h"
#include "zl_xxh64.h"

#define ZL_GROUP_CHUNKS 8192u  // chunks encoded per launch group (1 GiB of input); bounds the scratch

struct zl_cctx_s {
    int level = 3;
    int checksum = 0;
    int nb_workers = 0;
    int layout = 0;  // ZL_c_frameLayout: 0 auto, 1 independent frames, 2 one frame
    int pipeline = 1;  // ZL_c_pipeline: host entries overlap upload, kernels and download for large inputs
    u32 chunk_cap = ZL_CHUNK;  // the largest src_size in the chunk table being enqueued, if the caller knows it (zl_compress_batch): sizes the per-chunk scratch
    bool any_small = false;  // the chunk table being enqueued holds inputs of at most one chunk (zl_chunk_desc::flags bit 0)
    bool one_input = false;  // the chunk table being enqueued cuts ONE contiguous input, in order (what the deep match finder of levels >= 6 sorts)
    u64 input_total = 0;     // ... bytes of that input
    // the deep match finder's scratch (a 64-bit word per input byte of a launch group, two key arrays and the sort's storage per
    // section) did not fit, or the sort refused: deep_failed says so to the entry that made the call, which sets deep_denied, gives the deep
    // buffers back and makes the call again on the table parse at its deepest setting -- levels >= 6 degrade instead of failing.  The denial
    // lasts for the streaming session (a frame's header names ONE window), one-shot calls ask again every time.
    bool deep_failed = false, deep_denied = false;
    // ZL_c_faultInjection (tests): bit 0 -- every allocation of the deep finder "fails"; bits 4 and up: k -- from the k-th full job of a streaming
    // session on, the job's thread "cannot be had" (jobs_started counts them; reset with the session)
    int fault = 0, jobs_started = 0;
    int last_slots = 1;  // device slots the last zl_compress2 call used
    int devices = 1;     // ZL_c_devices: device slots zl_compress2 spreads a large input over (0: every visible device)
    std::vector<zl_cctx_s*> peers;  // one context per slot, created on first use
    int seek_table = 0;  // ZL_c_seekTable: zl_compress2 appends the skippable seek-table frame (zl_format.h)
    std::vector<u8> dict;
    bool dict_uploaded = false, dict_formatted = false;
    u32 dict_id = 0, dict_content_off = 0, dict_use = 0;  // dict_use: trailing content bytes the matcher can reach
    DevBuf d_dict, d_dict_tables, d_dict_hash, d_cdict;
    hipStream_t own_stream = nullptr;
    DevBuf d_src, d_dst, d_chunks, d_fsize, d_foff, d_totals;
    // per-launch-group scratch.  Set 0 serves the one-stream paths; the host entries' pipeline (zl_compress2 of a large
    // input) keeps ZL_PIPE_SETS groups in flight, each on its own stream with its own set.
    struct EncSet {
        DevBuf lits, seqs, sst, meta, staged, far_table, far_hints, far_lists, deep_best, deep_scratch, deep_chain, order, handover, sub_lits, sub_seqs, sub_meta;
        hipStream_t st = nullptr;
        void release() { for (DevBuf* b : {&lits, &seqs, &sst, &meta, &staged, &far_table, &far_hints, &far_lists, &deep_best, &deep_scratch, &deep_chain, &order, &handover, &sub_lits, &sub_seqs, &sub_meta}) b->release(); if (st) (void)hipStreamDestroy(st); st = nullptr; }
    };
    EncSet sets[3];
    hipStream_t copy_stream = nullptr, down_stream = nullptr;  // the process's pair for this context's device (zl_copy_pair.h): not owned
    double batch_ms[5] = {0, 0, 0, 0, 0};  // ZL_c_profile: the last zl_compress_batch call: {pack on the host, upload, kernels, download, unpack on the host}
    int copy_pair_duplex = -1;  // -1: not bound yet; 1 / 0: the pair's copies overlap / take turns (zl_bind_copy_pair)
    std::vector<hipEvent_t> pipe_events;  // grow-only pool for the pipeline
    PinBuf h_chunks, h_fsize, h_totals, h_stage_in, h_stage_out;
    // pending device-resident call
    bool pending = false;
    hipStream_t pending_stream = nullptr;
    size_t pending_chunks = 0, pending_groups = 0;
    bool pending_hash = false;        // the pending call writes ONE frame with a content checksum: XXH64 of the whole input goes behind the last block
    const u8* pending_src = nullptr;  // ... the input (device) and its size, where the frame is being written
    size_t pending_footer = 0;        // ... bytes zl_compress_device_async put behind the frame itself (the checksum)
    u8* pending_dst = nullptr;
    size_t pending_n = 0;
    // diagnostics
    int profile = 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_far = nullptr;   // start of the far-table passes (windowed layout)
    bool far_valid = false;
    double far_ms = 0;
    bool ev_valid = false;
    bool acc_valid = false;          // ... and the sums over a call's launch groups, when it had several (the deep levels)
    double acc_ms[4] = {0, 0, 0, 0}, acc_far = 0;
    double kernel_ms[4] = {0, 0, 0, 0};
    DevBuf d_stats;
    u64 stats[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // streaming session (zl_compressStream2)
    struct {
        ByteBuf stage;  // input collected for the current job
        ByteBuf out;    // compressed bytes not handed out yet
        size_t out_pos = 0;
        u64 pledged = ZL_CONTENTSIZE_UNKNOWN, consumed = 0;
        bool started = false;   // input has been taken in this session
        bool emitted = false;   // a frame has been written in this session
        bool finishing = false; // ZL_e_end has compressed the rest; only handing out remains
        bool joined = false;    // the session's jobs are consecutive runs of blocks of ONE frame (decided when the first input arrives)
        bool flush_told = false;  // a ZL_e_flush has reported the input that stays collected since input last arrived
        zlx::Xxh64 hash;        // joined session with ZL_c_checksumFlag: the content checksum runs through all jobs
        // round 4: a full job runs on a thread of its own while the caller's next pieces are collected; every other entry that touches the
        // context waits for the threads first (cs_wait).  Round 6: TWO jobs in flight -- a job is an upload, a few kernels and a download, one
        // behind the other (2.6 ms for 64 MiB), and the link carries both directions at once: the second job's upload runs beside the first
        // one's kernels and download.  Each slot has a WORKER context of its own (device buffers, scratch sets, streams); the jobs' bytes are
        // handed out in the order the jobs were started.  (One at a time where ONE content checksum runs through all jobs of a frame.)
        struct Job {
            std::thread th;
            ByteBuf in, out;   // the job's input and what it wrote
            size_t ret = 0;    // ... its byte count or error code
        } job[2];
        int job_head = 0, job_count = 0;  // the oldest job not yet collected; jobs started and not yet collected
        zl_cctx_s* worker[2] = {nullptr, nullptr};
    } cs;
};

namespace {

// a streaming job that runs beside the caller (zl_compressStream2) is finished before anything else uses the context
inline void cs_wait(zl_cctx* c) {
    for (auto& j : c->cs.job) if (j.th.joinable()) j.th.join();
}

hipStream_t cctx_stream(zl_cctx* c, void* user) {
    if (user) return (hipStream_t)user;
    if (!c->own_stream) (void)hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    return c->own_stream;
}

inline size_t chunks_of(size_t n) { return n == 0 ? 1 : (n + ZL_CHUNK - 1) / ZL_CHUNK; }

// Levels 6 - 15: the deep match finder (zl_deep.hip) with the reference's own search depth, lazy depth and window at that level
// (zstd.c:29523-29550: searchLog 3 / 4 / 4 / 4 / 5 / 6 / 6 at levels 6 - 12 -> 2^searchLog candidates; lazy at 6 - 7, lazy2 from 8;
// windowLog 21 up to level 8, 22 from 9).  Levels 13 - 15 (btlazy2: a binary tree finds the longest match among more candidates
// than a chain visits) search deeper still.  Levels 16 - 22 (round 6): 15's search inside the reference's window at that level, every
// position's CHAIN kept beside its best candidate, 15's lazy parse to have statistics, and then the price-based parse of zl_opt.hip
// (`opt` passes of it: the reference's btopt / btultra / btultra2, zstd.c:34586-36163).
struct DeepLevel { u32 k, lazy2, wlog, opt; };
// Levels 3 - 5 on an input of ZL_DEEP_SMALL_MIN .. ZL_DEEP_SMALL_MAX bytes search the same way, a few candidates deep (round 5): a call
// that small is bound by its launches, not by its kernels -- the sort and the search of 1 MiB cost less than the far passes and the
// table parse do (python sources, 1 MiB, host to host: 2.7 ms by the tables, 2.2 ms by the deep finder), and what it writes is
// 4 - 8 % SMALLER than the reference's at the same level where the table parse wrote 8 - 12 % more.  Larger inputs (where the table
// parse's 100 GB/s is the product) and the smallest (one sparse table pass, 0.15 ms a call) keep the table parse.
#ifndef ZL_DEEP_SMALL_MIN
#define ZL_DEEP_SMALL_MIN (16u << 10)
#endif
#ifndef ZL_DEEP_SMALL_MAX
#define ZL_DEEP_SMALL_MAX (16u << 20)
#endif
#ifndef ZL_DEEP_SUB
#define ZL_DEEP_SUB 8u  // workgroups that share a chunk in the deep parse of a small launch (a power of two; 1: never).  16 and 32 were measured: 6 - 9 % off a small call's
                        // time, and config C1's buffer repeated to 1 MiB 1.098 -> 1.174x the reference's size (every part starts without a repeat offset and probes for itself)
#endif
#ifndef ZL_BT_K2
#define ZL_BT_K2 256u     // levels 13 - 15 (the reference's btlazy2: a binary tree finds the longest match): candidates of a second search pass over an eight-byte hash whose best replaces the first pass's where it gains more (16 MiB of the synthetic text at level 15: 1.095 -> 1.039 of the reference's size, the python sources 1.012 -> 1.006, for 8 -> 14 ms of the call); 0: one pass
#endif
#ifndef ZL_OPT_K2
#define ZL_OPT_K2 256u    // levels 16 - 22: candidates per position of the second search pass (eight-byte hash; zl_deep.hip); 0: one pass
#endif
#ifndef ZL_OPT_PASSES
#define ZL_OPT_PASSES 2u  // price-based passes at levels 16 - 22 (the model, python sources at level 16: 1.011 / 1.002 / 1.000 of the reference's size with one / two / three; 0: level 15's parse, as round 5)
#endif
inline bool deep_small(int level, u64 total) { return level >= 3 && level <= 5 && total > ZL_DEEP_SMALL_MIN && total <= ZL_DEEP_SMALL_MAX; }
inline DeepLevel deep_level(int level, u64 total) {
    static const DeepLevel tab[17] = {{8, 0, 21, 0}, {16, 0, 21, 0}, {16, 1, 21, 0}, {16, 1, 22, 0}, {32, 1, 22, 0}, {64, 1, 22, 0}, {96, 1, 22, 0}, {128, 1, 22, 0}, {192, 1, 22, 0}, {256, 1, 22, 0},
                                      // 16 - 22: level 15's search and parse inside the reference's WINDOW at that level (zstd.c:29540-29550: windowLog 22 / 23 / 23 / 23 / 25 / 26 / 27;
                                      // here at most a section, 2^26).  The sort hands a position its K most recent twins whatever the distance, so the window costs nothing
                                      {256, 1, 22, ZL_OPT_PASSES}, {256, 1, 23, ZL_OPT_PASSES}, {256, 1, 23, ZL_OPT_PASSES}, {256, 1, 23, ZL_OPT_PASSES}, {256, 1, 25, ZL_OPT_PASSES}, {256, 1, 26, ZL_OPT_PASSES}, {256, 1, 26, ZL_OPT_PASSES}};
    static const DeepLevel small[3] = {{3, 0, 21, 0}, {4, 0, 21, 0}, {6, 0, 21, 0}};
    if (deep_small(level, total)) return small[level - 3];
    return tab[std::min(std::max(level, 6), 22) - 6];
}
// ... which needs ONE contiguous input and no dictionary (a dictionary's content is not part of what is sorted): batches of
// objects and dictionary contexts at these levels run the table-based parse at its deepest setting instead
inline bool deep_on(const zl_cctx* c) { return (c->level >= 6 || deep_small(c->level, c->input_total)) && c->one_input && c->dict_use == 0 && !c->deep_denied; }
// what a denied deep call leaves behind goes back to the device before the call is made again
inline void deep_release(zl_cctx* c) {
    for (auto& es : c->sets) for (DevBuf* b : {&es.deep_best, &es.deep_scratch, &es.deep_chain, &es.sub_lits, &es.sub_seqs, &es.sub_meta}) b->release();
}
// Runs `call` and, when the deep finder could not get what it needs, once more without it.
template <class F>
inline size_t with_deep_fallback(zl_cctx* c, F&& call) {
    c->deep_failed = false;
    size_t r = call();
    if (ZL_IS_ERR(r) && c->deep_failed && !c->deep_denied) {
        (void)hipDeviceSynchronize();  // (launch groups in front of the one that failed may still be running on the buffers given back below)
        c->deep_denied = true;
        c->deep_failed = false;
        deep_release(c);
        for (zl_cctx* p : c->peers) { p->deep_denied = true; deep_release(p); }
        r = call();
    }
    return r;
}

inline u32 zl_rd32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

// upload + digest the dictionary once per context: where its content starts (device parse of the
// entropy section, shared with the decoder) and the match tables as they look after its last byte
size_t cctx_prepare_dict(zl_cctx* c, hipStream_t st) {
    if (c->dict.empty() || c->dict_uploaded) return 0;
    if (!c->d_dict.ensure(c->dict.size() + 16)) return kGpuUnavailable;
    if (hipMemcpyAsync(c->d_dict.p, c->dict.data(), c->dict.size(), hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    c->dict_content_off = 0;
    if (c->dict_formatted) {
        if (!c->d_dict_tables.ensure(sizeof(zl_dict_tables))) return kGpuUnavailable;
        if (zl_launch_digest_dict((const u8*)c->d_dict.p, (u32)c->dict.size(), c->d_dict_tables.p, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        zl_dict_tables hdr;
        if (hipMemcpyAsync(&hdr, c->d_dict_tables.p, sizeof(hdr), hipMemcpyDeviceToHost, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (hdr.status) return ZL_ERR(hdr.status);
        c->dict_content_off = hdr.content_off;
        if (!c->d_cdict.ensure(ZL_CDICT_BYTES)) return kGpuUnavailable;
        if (zl_launch_cdict_build(c->d_dict_tables.p, c->d_cdict.p, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    }
    const size_t content = c->dict.size() - c->dict_content_off;
    c->dict_use = (u32)std::min<size_t>(content, ZL_DICT_REACH);
    if (c->dict_use >= 8) {
        if (!c->d_dict_hash.ensure(ZL_MATCH_TABLE_BYTES)) return kGpuUnavailable;
        const u8* tail = (const u8*)c->d_dict.p + c->dict.size() - c->dict_use;
        if (zl_launch_dict_hash(tail, c->dict_use, (u32*)c->d_dict_hash.p, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    } else {
        c->dict_use = 0;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    c->dict_uploaded = true;
    return 0;
}

// The per-chunk scratch strides for a chunk table whose largest entry holds `cap` bytes.  Full chunks: the constants of
// zl_encode.h.  A batch of small objects (config C4: 100k objects of a few hundred bytes) gets strides to match, so that
// the whole batch is ONE launch group instead of thirteen (each group's four kernels and its size scan drain the device).
struct EncStrides { u32 lit, seq, out, group; };
inline EncStrides enc_strides(u32 cap) {
    EncStrides s;
    if (cap >= ZL_CHUNK) { s.lit = ZL_LIT_STRIDE_ENC; s.seq = ZL_SEQ_STRIDE; s.out = ZL_OUT_STRIDE; s.group = ZL_GROUP_CHUNKS; return s; }
    cap = std::max<u32>(cap, 1024u);
    s.lit = (cap + 64 + 15) & ~15u;
    s.seq = (cap / 4 + 8 + 7) & ~7u;                      // (min match 4; the code arrays of zl_entropy_kernel want a multiple of 8)
    s.out = (cap + 1024 + 15) & ~15u;                     // (the entropy kernel keeps 512 bytes of headroom for a sequences header while it builds a frame)
    const size_t per = (size_t)s.lit + (size_t)s.seq * (sizeof(zl_seq) + ZL_SST_BYTES) + s.out;
    s.group = (u32)std::max<size_t>(ZL_GROUP_CHUNKS, std::min<size_t>(((size_t)8 << 30) / per, 1u << 20));  // scratch of a group: at most what 8192 full chunks take
    return s;
}

// One launch group: far passes (windowed layout) -> E1 -> E2 -> size scan -> gather for chunks [first, first + cnt) on
// stream st, with scratch set `es`.  The group's frames land behind those of group g - 1 in d_dst: its size scan starts
// from d_totals[g] (wait_prev: an event behind which that word is valid, when the groups run on different streams) and
// leaves d_totals[g + 1]; scanned: recorded behind the scan, if given.
// frame_chunk0: where the chunk table's entry 0 sits inside the frame (non-zero for a device's shard of an input).
size_t cctx_enqueue_group(zl_cctx* c, zl_cctx_s::EncSet& es, u8* d_dst, const u8* d_src, size_t first, u32 cnt, size_t nchunks, size_t g, hipStream_t st,
                          u64 joined_total, bool prof, hipEvent_t wait_prev, hipEvent_t scanned, size_t frame_chunk0 = 0) {
    const bool joined = joined_total != 0;
    const EncStrides sd = enc_strides(c->chunk_cap);
#ifdef ZL_DEV_SWITCHES
    static const bool dbg_sync = getenv("ZL_DEBUG_SYNC") != nullptr;  // (development: drain the stream behind every kernel and say which one it was)
    auto stage_ok = [&](const char* what) { if (dbg_sync) { const hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[zl enc] group %zu first %zu cnt %u: %s -> %d\n", g, first, cnt, what, (int)e); } };
#else
    auto stage_ok = [](const char*) {};
#endif
    if (!es.meta.ensure((size_t)cnt * sizeof(zl_chunk_meta)) || !es.lits.ensure((size_t)cnt * sd.lit) || !es.seqs.ensure((size_t)cnt * sd.seq * sizeof(zl_seq)) ||
        !es.sst.ensure((size_t)cnt * sd.seq * ZL_SST_BYTES + 64) || !es.staged.ensure((size_t)cnt * sd.out))
        return kGpuUnavailable;
    const bool deep = deep_on(c);
    const DeepLevel dl = deep_level(c->level, c->input_total);
    if (joined && !deep && (!es.far_table.ensure(ZL_FAR_TABLE_BYTES(cnt)) || !es.far_hints.ensure((size_t)cnt * ZL_FAR_HINT_WORDS * 4) || !es.far_lists.ensure(ZL_FAR_LIST_BYTES(cnt)))) return kGpuUnavailable;
    c->far_valid = false;
    u64 deep_lo = 0;
    if (deep) {
        // every position's best earlier occurrence, section by section (zl_deep.hip), in front of the parse
        const zl_chunk_desc* ht = (const zl_chunk_desc*)c->h_chunks.p + first;
        zl_deep_args d;
        d.src = d_src;
        d.lo = deep_lo = ht[0].src_off;
        d.hi = ht[cnt - 1].src_off + ht[cnt - 1].src_size;
        d.total = c->input_total;
        d.k = dl.k;
        d.window = (1u << dl.wlog) - 1u;
        d.mml = (c->level < 6 || c->input_total > ((u64)256 << 10)) ? 5u : 4u;  // (zstd.c:29523-29600: minMatch 5 above 256 KiB, 4 below, at levels 6 and up; 5 throughout at 3 - 5)
        d.frames = joined ? 0u : 1u;
        size_t scratch_bytes = zl_deep_scratch_bytes(d.hi - d.lo);  // (0: the sort could not say what it needs)
        if (scratch_bytes == 0 || (c->fault & 1) || !es.deep_best.ensure((size_t)(d.hi - d.lo) * 8 + 64) || !es.deep_scratch.ensure(scratch_bytes) ||
            (dl.opt && !es.deep_chain.ensure((size_t)(d.hi - d.lo) * 8 * ZL_DEEP_CHAIN + 64))) { c->deep_failed = true; return kGpuUnavailable; }
        d.chain = dl.opt ? (u64*)es.deep_chain.p : nullptr;
        d.k2 = dl.opt ? ZL_OPT_K2 : ((c->level >= 13 && deep_level(c->level, c->input_total).k >= 96) ? ZL_BT_K2 : 0u);
        d.mml2 = 8;
        d.scratch = es.deep_scratch.p;
        d.scratch_bytes = scratch_bytes;
        d.best = (u64*)es.deep_best.p;
        if (prof) { (void)hipEventRecord(c->ev_far, st); c->far_valid = true; }
        hipError_t de = zl_launch_deep_search(&d, st);
        if (de == hipErrorOutOfMemory) {
            // the sort asked for more temporary storage at this section's size than its size class had said (zl_deep.hip sort_temp_bytes
            // remembers the larger figure): once more with that; the buffer is replaced behind everything in flight (hipFree waits)
            scratch_bytes = zl_deep_scratch_bytes(d.hi - d.lo);
            if (scratch_bytes == 0 || !es.deep_scratch.ensure(scratch_bytes)) { c->deep_failed = true; return kGpuUnavailable; }
            d.scratch = es.deep_scratch.p;
            d.scratch_bytes = scratch_bytes;
            de = zl_launch_deep_search(&d, st);
        }
        if (de != hipSuccess) { c->deep_failed = true; return ZL_ERR(ZL_E_GENERIC); }
        stage_ok("deep search");
    } else if (joined) {
        if (prof) { (void)hipEventRecord(c->ev_far, st); c->far_valid = true; }
        // the far table over this launch group's input, then the hint words the match kernel reads
        zl_far_args f;
        f.chunks = (const zl_chunk_desc*)c->d_chunks.p + first;
        f.src = d_src;
        f.total = joined_total;
        f.chunk_base = (u32)(frame_chunk0 + first);
        f.sample_log = c->level >= 5 ? 2u : (u32)ZL_FAR_SAMPLE_LOG;  // the operating points above level 3: denser far sampling
        f.table = (u32*)es.far_table.p;
        f.hints = (u32*)es.far_hints.p;
        f.lists = (u32*)es.far_lists.p;
        if (zl_launch_far_build(&f, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stage_ok("far build");
        if (zl_launch_far_probe(&f, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stage_ok("far probe");
    }
    // launch order: longest first (zl_launch_chunk_order), for launches that fill the device several times over
    const u32* order = nullptr;
    if (cnt >= 2048 && c->chunk_cap >= ZL_CHUNK && c->dict_use == 0) {  // (a dictionary frame's first chunk is a launch of its own)
        if (!es.order.ensure((size_t)cnt * 8 + 64)) return kGpuUnavailable;
        if (zl_launch_chunk_order((const zl_chunk_desc*)c->d_chunks.p + first, d_src, cnt, (u32*)es.order.p + cnt, (u32*)es.order.p, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        order = (const u32*)es.order.p;
        stage_ok("order");
    }
    zl_match_args m;
    m.chunks = (const zl_chunk_desc*)c->d_chunks.p + first;
    m.src = d_src;
    m.lits = (u8*)es.lits.p;
    m.seqs = (zl_seq*)es.seqs.p;
    m.meta = (zl_chunk_meta*)es.meta.p;
    m.lit_stride = sd.lit;
    m.seq_stride = sd.seq;
    m.stats = nullptr;
    m.sparse_only = c->level < 3 ? 1u : 0u;
    m.hi = c->level >= 4 ? 1u : 0u;  // (level 4, round 5: the larger tables and the longer verification without level 5's denser far search, second lazy step and repeat-offset history -- what level 5 was until then)
    m.lazy2 = c->level >= 5 ? 1u : 0u;  // (level 5 too since round 5: with the one-step rule alone its python sources were 1.051x the reference's level 5)
    m.dict = c->dict_use ? (const u8*)c->d_dict.p + c->dict.size() - c->dict_use : nullptr;
    m.dict_len = c->dict_use;
    m.dict_hash = c->dict_use ? (const u32*)c->d_dict_hash.p : nullptr;
    m.far_hints = (joined && !deep) ? (const u32*)es.far_hints.p : nullptr;
    m.intra = c->any_small ? 1u : 0u;
    m.accel = c->level < 0 ? (u32)std::min(-c->level, 7) : 0u;
    m.deep_best = deep ? (const u64*)es.deep_best.p : nullptr;
    m.deep_lo = deep_lo;
    m.total = c->input_total;
    m.joined = joined ? 1u : 0u;
    m.sub = 0;
    m.sub_lits = nullptr; m.sub_seqs = nullptr; m.sub_meta = nullptr;
    m.order = order;
    if (deep) {
        m.lazy2 = dl.lazy2;
        // a launch that would leave most of the device idle: several workgroups share a chunk (zl_match_deep_kernel)
        // (a function of the INPUT's size, not of the group's: every route a call can take -- one launch, sections through the
        // upload slots, a stream's jobs -- writes the same bytes)
        u32 parts = ZL_DEEP_SUB;
        while (parts > 1 && chunks_of(c->input_total) * parts > 2048) parts >>= 1;
        if (dl.opt) {  // the price-based parse of levels 16+ walks whole chunks, and writes beside the parse it prices itself by (zl_opt.hip): the parts' arrays
            if (!es.sub_lits.ensure((size_t)cnt * sd.lit) || !es.sub_seqs.ensure((size_t)cnt * sd.seq * sizeof(zl_seq))) { c->deep_failed = true; return kGpuUnavailable; }
            m.sub_lits = (u8*)es.sub_lits.p; m.sub_seqs = (zl_seq*)es.sub_seqs.p;
        } else
        if (parts > 1 && c->chunk_cap >= ZL_CHUNK) {
            if (!es.sub_lits.ensure((size_t)cnt * sd.lit) || !es.sub_seqs.ensure((size_t)cnt * sd.seq * sizeof(zl_seq)) || !es.sub_meta.ensure((size_t)cnt * parts * 8 + 64)) { c->deep_failed = true; return kGpuUnavailable; }
            m.sub = parts;
            m.sub_lits = (u8*)es.sub_lits.p; m.sub_seqs = (zl_seq*)es.sub_seqs.p; m.sub_meta = (u32*)es.sub_meta.p;
        }
    }
#ifdef ZL_TRACE
    // development build: the match kernel stamps a timeline of one workgroup's iterations behind the 128 bytes of phase sums
    if (prof && c->profile >= 2 && c->d_stats.ensure(128 + ZL_TRACE_BYTES)) {
        (void)hipMemsetAsync(c->d_stats.p, 0, 128 + ZL_TRACE_BYTES, st);
        m.stats = (u64*)c->d_stats.p;
    }
#else
    if (prof && c->profile >= 2 && c->d_stats.ensure(128)) {
        (void)hipMemsetAsync(c->d_stats.p, 0, 128, st);
        m.stats = (u64*)c->d_stats.p;
    }
#endif
    if (prof) (void)hipEventRecord(c->ev[0], st);
    if (deep) {
        if (zl_launch_match_deep(&m, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        // levels 16 - 22: the lazy parse's sequences seed the statistics of the price-based one, which replaces them (the best words' buffer,
        // which the lazy parse has read, takes the arrivals)
        if (dl.opt && zl_launch_opt_parse(&m, (const u64*)es.deep_chain.p, (u64*)es.deep_best.p, cnt, dl.opt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    } else if (joined && c->dict_use) {
        // ONE frame with a dictionary (zstd.c:27550: the dictionary's content is history in front of the frame's first byte):
        // the frame's first chunk is searched with the dictionary in front of it, exactly as a dictionary frame of its own
        // would be; the chunks behind it see the frame (far matches reach into chunk 0, not into the dictionary).
        u32 skip = 0;
        if (frame_chunk0 + first == 0) {
            zl_match_args m0 = m;
            m0.far_hints = nullptr;
            if (zl_launch_match(&m0, 1, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            skip = 1;
        }
        zl_match_args m1 = m;
        m1.chunks += skip; m1.lits += (size_t)skip * m.lit_stride; m1.seqs += (size_t)skip * m.seq_stride; m1.meta += skip;
        m1.far_hints += (size_t)skip * ZL_FAR_HINT_WORDS;
        m1.dict = nullptr; m1.dict_len = 0; m1.dict_hash = nullptr;
        if (zl_launch_match(&m1, cnt - skip, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    } else if (zl_launch_match(&m, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    stage_ok("match");
    if (prof) (void)hipEventRecord(c->ev[1], st);
    zl_entropy_args e;
    e.chunks = m.chunks;
    e.src = d_src;
    e.lits = m.lits;
    e.seqs = m.seqs;
    e.meta = m.meta;
    e.seq_states = (u8*)es.sst.p;
    e.out = (u8*)es.staged.p;
    e.frame_size = (u32*)c->d_fsize.p + first;
    e.lit_stride = sd.lit;
    e.seq_stride = sd.seq;
    e.out_stride = sd.out;
    e.checksum = (u32)c->checksum;
    e.dict_id = c->dict_use ? c->dict_id : 0;  // (a raw-content dictionary has no id: zstd.c:27550)
    e.raw_lits = c->level < 0 ? 1u : 0u;
    e.exact_rep = (c->level >= 5 || deep) ? 1u : 0u;
    e.joined = joined ? 1u : 0u;
    e.chunk_base = (u32)(frame_chunk0 + first);
    e.total_chunks = (u32)nchunks;
    e.wlog = deep ? dl.wlog : (u32)ZL_FAR_WLOG;
    e.total_size = joined_total;
    e.cdict = (c->dict_use && c->dict_formatted && !joined) ? c->d_cdict.p : nullptr;
    e.order = order;
#ifndef ZL_ENT_SMALL
#define ZL_ENT_SMALL 1  // 0: every object through zl_entropy_kernel, as rounds 1-5
#endif
#ifndef ZL_ENT_SPLIT
#define ZL_ENT_SPLIT 0  // 1: the entropy stage as two kernels with a register allocation each (zl_encode.hip zl_entropy_kernel<1>, <2>).  Measured, round 5, same bytes: 18 and 0 vector registers spilled in the halves (64 and 114 registers) against the one kernel's 60 -- and 1.98 ms per GiB of the mix against the one kernel's 1.75 (second half at 5 / 8 workgroups per CU: 1.88 / 2.03): the spills are not what the stage waits for
#endif
    e.handover = nullptr;
    // small objects on a dictionary's entropy tables (zl_encode.hip zl_entropy_small_kernel): levels 3 and 4 (repeat offset 1 only), no checksum, literals Huffman-coded.
    // (Whatever the call's other objects are like -- until the end of round 6 only calls whose LARGEST object was below 16 KiB: the kernel takes a valid
    // table as it is where the general kernel weighs it against the predefined one, so an object came out of a batch of small ones with other bytes --
    // of the same size, scripts/fuzz_multi.py seed 162 -- than out of the one-object entry or a batch that held a large object too.)
    e.small_done = (ZL_ENT_SMALL && e.cdict != nullptr && !e.exact_rep && !e.raw_lits && !e.checksum && !e.joined && e.order == nullptr) ? 1u : 0u;
    if (ZL_ENT_SPLIT && c->profile < 2) {  // (the in-kernel phase stamps of ZL_c_profile = 2 describe the one-kernel form)
        if (!es.handover.ensure((size_t)cnt * ZL_ENT_HANDOVER + 64)) return kGpuUnavailable;
        e.handover = (u8*)es.handover.p;
    }
    e.stats = m.stats ? m.stats + 8 : nullptr;
    if (zl_launch_entropy(&e, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    stage_ok("entropy");
    if (prof) (void)hipEventRecord(c->ev[2], st);
    u64* totals = (u64*)c->d_totals.p;
    if (wait_prev && hipStreamWaitEvent(st, wait_prev, 0) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (zl_launch_scan_sizes(e.frame_size, (u64*)c->d_foff.p + first, totals + g, totals + g + 1, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (scanned && hipEventRecord(scanned, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (prof) (void)hipEventRecord(c->ev[3], st);
    zl_gather_args ga;
    ga.staged = e.out;
    ga.frame_size = e.frame_size;
    ga.frame_off = (const u64*)c->d_foff.p + first;
    ga.dst = d_dst;
    ga.out_stride = sd.out;
    if (zl_launch_gather(&ga, cnt, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    stage_ok("scan + gather");
    if (prof) { (void)hipEventRecord(c->ev[4], st); c->ev_valid = true; }
    return 0;
}

// Enqueue the whole encoder over a chunk table already written to c->h_chunks (nchunks entries), one launch group
// after the other on one stream.  Frames end up back to back in d_dst; per-chunk sizes in d_fsize, per-group running
// totals in d_totals[g+1].  joined_total != 0: the chunks are the consecutive pieces of ONE input of that many bytes
// and become one frame (windowed layout: far matches, one header, one Last_Block).
size_t cctx_enqueue(zl_cctx* c, u8* d_dst, const u8* d_src, size_t nchunks, hipStream_t st, u64 joined_total = 0) {
    { const size_t pd = cctx_prepare_dict(c, st); if (ZL_IS_ERR(pd)) return pd; }
    const bool joined = joined_total != 0;
    // (levels >= 6: the deep match finder keeps a 64-bit word per input byte of a launch group -- 8 GiB for a group of 1 GiB.  The group
    // shrinks, down to two sections, until that fits into half of what the device has free: the bytes do not depend on the grouping
    // (tests/test_fullsize_gpu.py), only the schedule does -- a launch over everything lets the expensive chunks of one kind of data
    // run beside the cheap ones of another: the 1 GiB mix at level 9 as ONE group: parse 7.7 ms, as four groups of 256 MiB: 10.6)
    size_t group = enc_strides(c->chunk_cap).group;
    if (deep_on(c) && nchunks > 2 * ((size_t)1 << ZL_FAR_SECTION_LOG)) {  // (a call of two sections or fewer is one group whatever is free: no question to ask the driver)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        // what a group of g chunks asks of the device: the candidate words (8 bytes per input byte), the per-chunk scratch of the parse and
        // the entropy stage (enc_strides), and -- whatever the group -- one section's two key arrays and the sort's storage (the
        // last two were not counted); what the set already holds counts as available to it
        const zl_cctx_s::EncSet& s0 = c->sets[0];
        const size_t have = free_b + s0.deep_best.cap + s0.deep_scratch.cap + s0.deep_chain.cap + s0.lits.cap + s0.seqs.cap + s0.sst.cap + s0.staged.cap;
        const EncStrides sd = enc_strides(c->chunk_cap);
        const size_t per_chunk = (size_t)ZL_CHUNK * 8 * (1 + (deep_level(c->level, c->input_total).opt ? ZL_DEEP_CHAIN : 0)) + sd.lit + (size_t)sd.seq * (sizeof(zl_seq) + ZL_SST_BYTES) + sd.out + sizeof(zl_chunk_meta);
        const size_t fixed = zl_deep_scratch_bytes((u64)1 << 40);
        while (group > ((size_t)1 << ZL_FAR_SECTION_LOG) && std::min(group, nchunks) * per_chunk + fixed > have / 2) group /= 2;
    }
    const size_t ngroups = (nchunks + group - 1) / group;
    if (!c->d_chunks.ensure(nchunks * sizeof(zl_chunk_desc)) || !c->d_fsize.ensure(nchunks * sizeof(u32)) ||
        !c->d_foff.ensure(nchunks * sizeof(u64)) || !c->d_totals.ensure((ngroups + 1) * sizeof(u64)))
        return kGpuUnavailable;
    if (hipMemcpyAsync(c->d_chunks.p, c->h_chunks.p, nchunks * sizeof(zl_chunk_desc), hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (hipMemsetAsync(c->d_totals.p, 0, sizeof(u64), st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    // ZL_c_profile: per-kernel events around ONE launch group; a call of several groups (the deep levels: four sections each) drains the
    // stream behind every group and adds the groups' times up (a diagnostic: the drains cost a few microseconds each)
    const bool prof = c->profile && (ngroups == 1 || deep_on(c));
    if (prof && !c->ev[0]) for (auto& e : c->ev) (void)hipEventCreate(&e);
    if (prof && (joined || deep_on(c)) && !c->ev_far) (void)hipEventCreate(&c->ev_far);  // (the start of the far passes, or of the deep search)
    c->ev_valid = false;
    c->acc_valid = false;
    for (int k = 0; k < 4; k++) c->acc_ms[k] = 0;
    c->acc_far = 0;
    for (size_t g = 0; g < ngroups; g++) {
        const size_t first = g * group;
        const u32 cnt = (u32)std::min<size_t>(group, nchunks - first);
        const size_t e = cctx_enqueue_group(c, c->sets[0], d_dst, d_src, first, cnt, nchunks, g, st, joined_total, prof, nullptr, nullptr);
        if (ZL_IS_ERR(e)) return e;
        if (prof && ngroups > 1) {
            if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            for (int k = 0; k < 4; k++) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]); c->acc_ms[k] += ms; }
            if (c->far_valid) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev_far, c->ev[0]); c->acc_far += ms; }
            c->acc_valid = true;
        }
    }
    if (!c->h_fsize.ensure(nchunks * sizeof(u32)) || !c->h_totals.ensure((ngroups + 1) * sizeof(u64))) return kGpuUnavailable;
    if (hipMemcpyAsync(c->h_fsize.p, c->d_fsize.p, nchunks * sizeof(u32), hipMemcpyDeviceToHost, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (hipMemcpyAsync(c->h_totals.p, c->d_totals.p, (ngroups + 1) * sizeof(u64), hipMemcpyDeviceToHost, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    c->pending_chunks = nchunks;
    c->pending_groups = ngroups;
    return 0;
}

// Which layout a call over n bytes of ONE input takes.  host_entry: the calls that stand in for ZSTD_compress2 /
// ZSTD_compressStream2, whose readers take the FIRST frame only (src/raw-file.c:150-155) -- there one frame is the
// default, with or without a checksum or a dictionary (round 2 wrote independent frames for those two: a first-frame
// reader then got 128 KiB back, silently).  The content checksum of one frame is one XXH64 over all of it
// (zstd.c:27733): computed on the host beside the upload (zl_xxh64.h).  A dictionary is history in front of the frame's
// first chunk (cctx_enqueue_group).
bool use_one_frame(const zl_cctx* c, size_t n, bool host_entry) {
    if (n <= ZL_CHUNK) return false;
    // (a dictionary frame of 4 GiB and more would meet neither the seek-table path nor the runs of blocks of the decoder, both of
    // which refuse a dictionary, and end on the general kernel's one workgroup: such input keeps independent frames -- ADVICE r3)
    if (!c->dict.empty() && n >= (size_t)0xFFFFD000ull && c->layout != 2) return false;
    if (c->layout == 1) return false;
    if (c->layout == 2) return true;
    return host_entry;
}

// Does this call append the seek table?  On request (ZL_c_seekTable) -- and, unasked, behind ONE frame of 4 GiB or more: the
// large-frame decoder plans frames below 4 GiB (32-bit positions, zl_decode3.hip), a larger one it could only hand to the
// general kernel, one workgroup for the whole frame: 4.5 GiB decoded in 400 s (scripts/huge_roundtrip.py).  With the table
// the same frame is decoded section by section (27 - 33 GB/s from host memory).  The table is a skippable frame behind the data
// frame: a reader that takes the first frame gets all of the content, as without it.
inline bool wants_seek_table(const zl_cctx* c, size_t n, bool one_frame) {
    return c->seek_table || (one_frame && n >= (size_t)0xFFFFD000ull);
}

// the seek table behind `total` bytes of output (zl_format.h), from the per-chunk sizes in c->h_fsize: zstd's seekable format
// for independent frames, the private trailer for the runs of blocks of one frame
size_t append_seek_table(const zl_cctx* c, u8* dst, size_t cap, size_t total, size_t nchunks, u64 content, bool one_frame) {
    const size_t need = one_frame ? ZL_INDEX_BYTES(nchunks) : ZL_SEEKABLE_BYTES(nchunks);
    if (nchunks > 0x1FFFFFFFu) return total;
    if (total + need > cap) return ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);
    u8* p = dst + total;
    auto w32 = [&](u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); p[3] = (u8)(v >> 24); p += 4; };
    w32(ZL_MAGIC_INDEX);
    w32((u32)(need - 8));
    const u32* fs = (const u32*)c->h_fsize.p;
    if (!one_frame) {
        for (size_t i = 0; i < nchunks; i++) {
            w32(fs[i]);
            w32((u32)std::min<u64>(ZL_CHUNK, content - (u64)i * ZL_CHUNK));
        }
        w32((u32)nchunks);
        *p++ = 0;  // Seek_Table_Descriptor: no per-frame checksums
        w32(ZL_SEEKABLE_MAGIC);
        return total + need;
    }
    for (size_t i = 0; i < nchunks; i++) w32(fs[i]);
    w32((u32)content); w32((u32)(content >> 32));
    w32((u32)nchunks);
    w32(1u);
    w32(ZL_INDEX_TAG);
    return total + need;
}

// the frame's content checksum behind its last block (zstd.c:27733 ZSTD_writeEpilogue): low 32 bits of XXH64, little-endian
inline size_t append_checksum(u8* dst, size_t cap, size_t total, u32 x) {
    if (total + 4 > cap) return ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);
    dst[total] = (u8)x; dst[total + 1] = (u8)(x >> 8); dst[total + 2] = (u8)(x >> 16); dst[total + 3] = (u8)(x >> 24);
    return total + 4;
}

size_t fill_chunk_table(zl_cctx* c, size_t n) {
    const size_t nchunks = chunks_of(n);
    if (!c->h_chunks.ensure(nchunks * sizeof(zl_chunk_desc))) return kGpuUnavailable;
    zl_chunk_desc* t = (zl_chunk_desc*)c->h_chunks.p;
    for (size_t i = 0; i < nchunks; i++) {
        t[i].src_off = (u64)i * ZL_CHUNK;
        t[i].src_size = (u32)std::min<size_t>(ZL_CHUNK, n - i * (size_t)ZL_CHUNK);
        t[i].flags = nchunks == 1 ? 1u : 0u;
    }
    c->any_small = nchunks == 1;
    c->one_input = true;
    c->input_total = n;
    return nchunks;
}

}  // namespace

extern "C" {

// One frame per chunk: header <= 13 bytes, block header 3, checksum 4; a chunk that does not
// compress is stored raw.  Always >= ZSTD_COMPRESSBOUND(n) (zstd.h:231) so a buffer sized by
// this bound is also large enough for what the reference would have produced.
size_t zl_compressBound(size_t n) {
    const size_t zb = n + (n >> 8) + (n < ZL_BLOCK_MAX ? ((ZL_BLOCK_MAX - n) >> 11) : 0);
    return zb + chunks_of(n) * ZL_FRAME_OVERHEAD + ZL_SEEK_ROOM(chunks_of(n));  // (room for the optional seek table, ZL_c_seekTable)
}

size_t zl_compress_device_nframes(size_t n) { return chunks_of(n); }

zl_cctx* zl_createCCtx(void) { return new (std::nothrow) zl_cctx_s(); }

size_t zl_freeCCtx(zl_cctx* c) {
    if (!c) return 0;
    cs_wait(c);
    if (c->pending) (void)hipStreamSynchronize(c->pending_stream);
    if (c->ev_far) (void)hipEventDestroy(c->ev_far);
    for (zl_cctx* p : c->peers) (void)zl_freeCCtx(p);
    c->peers.clear();
    for (auto& w : c->cs.worker) { if (w) (void)zl_freeCCtx(w); w = nullptr; }
    for (auto& es : c->sets) es.release();
    for (hipEvent_t e : c->pipe_events) (void)hipEventDestroy(e);
    DevBuf* bufs[] = {&c->d_src, &c->d_dst, &c->d_chunks, &c->d_dict, &c->d_dict_tables, &c->d_dict_hash, &c->d_cdict, &c->d_fsize, &c->d_foff, &c->d_totals, &c->d_stats};
    for (DevBuf* b : bufs) b->release();
    PinBuf* pins[] = {&c->h_chunks, &c->h_fsize, &c->h_totals, &c->h_stage_in, &c->h_stage_out};
    for (PinBuf* b : pins) b->release();
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

// zstd.c:23223 ZSTD_CCtx_setParameter for the parameters the glue uses (src/cctx.c:72-95,265-284).
// The encoder has these operating points, and a context remembers the one it runs at, not the number it was asked for
// (zl_CCtx_getParameter and the host mirror's zstd_cctx_settings report that):
//   levels < 3   every other position probed throughout (cf. the step of ZSTD_fast, zstd.c:29340); reported as given
//   level 3      the double-hash parse with the dense body where the data is rich in matches (cf. ZSTD_dfast, zstd.c:29529);
//                in the one-frame layout one position in 8 takes part in the far search.
//   level 4      ... with match tables four times the size and verification up to 72 bytes per lane (the reference's level 4 is its
//                level 3 with larger tables too, zstd.c:29530)
//   level 5      level 4 with one position in 4 in the far search, the two-step lazy rule and the decoder's exact repeat-offset history;
//                its tables: four times the size (2^14 + 2^13 entries, three workgroups per CU instead
//                of seven), verification up to 72 bytes per lane and one position in 4 in the far search.
//   levels 6-15  the deep match finder (zl_deep.hip): every position's K most recent earlier occurrences inside the window,
//                K, the lazy depth and the window as the reference has them at that level (deep_level above); python sources in
//                one frame 0.996 / 0.996 / 0.989 of the reference's size at 6 / 9 / 12 (round 4's table-based point 6: 1.085 /
//                1.128 / 1.149).  Contexts with a dictionary and the batch entries run the table-based parse at its deepest
//                setting (two-step lazy rule, all three repeat offsets) at these levels.
//   levels 16-22 level 15's search and parse inside the reference's window at that level (8 MiB at 17 - 19, 32 MiB at 20, 64 MiB -- a
//                section -- at 21 and 22); the optimal parsers (zstd.c:34586-36163) are not built.
// Measured against the reference at the same level: within 5 % of its size at levels 1, 2, 3 and 6 ... 15
// (tests/test_compress_gpu.py).
size_t zl_CCtx_setParameter(zl_cctx* c, int param, int value) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    switch (param) {
    case ZL_c_compressionLevel:
        if (value < -131072 || value > 22) return ZL_ERR(ZL_E_PARAM_OUTOFBOUND);
        c->level = value > 22 ? 22 : (value == 0 ? 3 : value);  // (0: the default level, zstd.h:333 / zstd.c:29721; above 22: ZSTD_maxCLevel)
        return 0;
    case ZL_c_checksumFlag: c->checksum = value ? 1 : 0; return 0;
    case ZL_c_nbWorkers: c->nb_workers = value; return 0;
    case ZL_c_stableInBuffer:
    case ZL_c_stableOutBuffer: return 0;
    case ZL_c_profile: c->profile = value < 0 ? 0 : (value > 2 ? 2 : value); return 0;  // 1: per-kernel events; 2: also in-kernel phase stamps
    case ZL_c_frameLayout:
        if (value < 0 || value > 2) return ZL_ERR(ZL_E_PARAM_OUTOFBOUND);
        c->layout = value;
        return 0;
    case ZL_c_pipeline: c->pipeline = value ? 1 : 0; return 0;
    case ZL_c_seekTable: c->seek_table = value ? 1 : 0; return 0;
    case ZL_c_devices:
        if (value < 0 || value > 64) return ZL_ERR(ZL_E_PARAM_OUTOFBOUND);
        c->devices = value;
        return 0;
    case ZL_c_faultInjection:
        if (value < 0 || value > 0xFFFF) return ZL_ERR(ZL_E_PARAM_OUTOFBOUND);
        c->fault = value;
        c->jobs_started = 0;
        return 0;
    default: return ZL_ERR(ZL_E_PARAM_UNSUPPORTED);
    }
}

size_t zl_CCtx_getParameter(const zl_cctx* c, int param, int* value) {
    if (!c || !value) return ZL_ERR(ZL_E_GENERIC);
    switch (param) {
    case ZL_c_compressionLevel: *value = c->level; return 0;
    case ZL_c_checksumFlag: *value = c->checksum; return 0;
    case ZL_c_nbWorkers: *value = c->nb_workers; return 0;
    case ZL_c_frameLayout: *value = c->layout; return 0;
    case ZL_c_seekTable: *value = c->seek_table; return 0;
    case ZL_c_devices: *value = c->devices; return 0;
    default: return ZL_ERR(ZL_E_PARAM_UNSUPPORTED);
    }
}

// zstd.c:23826 ZSTD_CCtx_loadDictionary: copied, digested on the device at first use.  The encoder
// uses the dictionary's CONTENT as history in front of every frame (zstd.c:27550
// ZSTD_loadZstdDictionary / the dfast dictionary variant zstd.c:30120): the match tables start out
// filled with its positions, offsets may reach into it, and the frame header names its id.  The
// dictionary's entropy tables are not reused yet (frames describe their own or use the predefined
// ones), so the ratio gain is the content's.
size_t zl_CCtx_loadDictionary(zl_cctx* c, const void* dict, size_t n) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    c->dict.clear();
    c->dict_uploaded = false;
    c->dict_formatted = false;
    c->dict_id = 0;
    c->dict_use = 0;
    if (dict && n) {
        c->dict.assign((const u8*)dict, (const u8*)dict + n);
        if (n >= 8 && zl_rd32((const u8*)dict) == ZL_MAGIC_DICT) {
            c->dict_formatted = true;
            c->dict_id = zl_rd32((const u8*)dict + 4);
        }
    }
    return 0;
}

// ONE frame with a content checksum through the device-resident entry: the XXH64 of the input and the four footer bytes.
// Runs inside zl_compress_device_async, on the calling thread, BEFORE it returns (round 3 did it in _finish: until then the frame
// in d_dst said "checksum" and had none, and a caller that overwrote d_src in stream order behind _async got a wrong one written
// -- ADVICE r3).  With this configuration the call is therefore not asynchronous: it returns once the frame is complete.
static size_t seal_one_frame_checksum(zl_cctx* c) {
    if (hipStreamSynchronize(c->pending_stream) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    size_t total = (size_t)((const u64*)c->h_totals.p)[c->pending_groups];
    {
        // ONE frame with a content checksum, content in device memory: the XXH64 chain runs on the host (zl_xxh64.h says why),
        // over pieces brought down through two pinned buffers so that the copy of one piece hides behind the hashing of the
        // one before.  Rare on this entry (its default layout is independent frames, each checked on the device).
        const size_t piece = (size_t)16 << 20;
        if (!c->h_stage_in.ensure(2 * piece)) return kGpuUnavailable;
        zlx::Xxh64 h;
        hipStream_t st = c->pending_stream;
        const size_t n = c->pending_n;
        size_t sent = 0, k = 0;
        auto fetch = [&](size_t slot) { const size_t len = std::min(piece, n - sent); const bool ok = hipMemcpyAsync((u8*)c->h_stage_in.p + slot * piece, c->pending_src + sent, len, hipMemcpyDeviceToHost, st) == hipSuccess; sent += len; return ok ? len : (size_t)0; };
        size_t len_cur = n ? fetch(0) : 0;
        while (len_cur) {
            if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            const size_t len_next = sent < n ? fetch((k + 1) & 1) : 0;
            h.update((const u8*)c->h_stage_in.p + (k & 1) * piece, len_cur);
            len_cur = len_next;
            k++;
        }
        if (h.total != n) return ZL_ERR(ZL_E_GENERIC);
        const u32 x = h.low32();
        u8 le[4] = {(u8)x, (u8)(x >> 8), (u8)(x >> 16), (u8)(x >> 24)};
        if (hipMemcpyAsync(c->pending_dst + total, le, 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);  // (the bound leaves room)
        ((u32*)c->h_fsize.p)[c->pending_chunks - 1] += 4;  // (the last chunk's size covers the frame's footer: what the decoder's chunk hint expects)
        c->pending_hash = false;
    }
    c->pending_footer = 4;
    return 0;
}

size_t zl_compress_device_async(zl_cctx* c, void* d_dst, size_t cap, const void* d_src, size_t n, void* stream) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    if (!device_ok()) return kGpuUnavailable;
    if (c->pending) return ZL_ERR(ZL_E_STAGE_WRONG);
    if (cap < zl_compressBound(n)) return ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);
    hipStream_t st = cctx_stream(c, stream);
    const size_t nchunks = fill_chunk_table(c, n);
    if (ZL_IS_ERR(nchunks)) return nchunks;
    const bool one = use_one_frame(c, n, false);
    if (!c->cs.started) c->deep_denied = false;
    const size_t e = with_deep_fallback(c, [&]() { return cctx_enqueue(c, (u8*)d_dst, (const u8*)d_src, nchunks, st, one ? (u64)n : 0); });
    if (ZL_IS_ERR(e)) return e;
    c->pending = true;
    c->pending_stream = st;
    c->pending_hash = one && c->checksum;
    c->pending_src = (const u8*)d_src;
    c->pending_dst = (u8*)d_dst;
    c->pending_n = n;
    c->pending_footer = 0;
    if (c->pending_hash) {
        const size_t r = seal_one_frame_checksum(c);
        if (ZL_IS_ERR(r)) { c->pending = false; return r; }
    }
    return 0;
}

size_t zl_compress_device_finish(zl_cctx* c, size_t* frame_sizes) {
    if (!c || !c->pending) return ZL_ERR(ZL_E_STAGE_WRONG);
    c->pending = false;
    if (hipStreamSynchronize(c->pending_stream) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (c->ev_valid) {
        for (int k = 0; k < 4; k++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]);
            c->kernel_ms[k] = ms;
        }
        c->far_ms = 0;
        if (c->far_valid) { float ms = 0; (void)hipEventElapsedTime(&ms, c->ev_far, c->ev[0]); c->far_ms = ms; }
        if (c->acc_valid) { for (int k = 0; k < 4; k++) c->kernel_ms[k] = c->acc_ms[k]; c->far_ms = c->acc_far; }  // (several launch groups: their sums)
        if (c->d_stats.p) (void)hipMemcpy(c->stats, c->d_stats.p, 128, hipMemcpyDeviceToHost);
    }
    size_t total = (size_t)((const u64*)c->h_totals.p)[c->pending_groups];
    total += c->pending_footer;
    if (frame_sizes) {
        const u32* fs = (const u32*)c->h_fsize.p;
        for (size_t i = 0; i < c->pending_chunks; i++) frame_sizes[i] = fs[i];
    }
    return total;
}

size_t zl_cctx_kernel_ms(const zl_cctx* c, double* ms) {
    if (!c || !ms || !c->profile || !c->ev_valid) return ZL_ERR(ZL_E_STAGE_WRONG);
    for (int k = 0; k < 4; k++) ms[k] = c->kernel_ms[k];
    return 0;
}
size_t zl_cctx_last_slots(const zl_cctx* c) { return c ? (size_t)c->last_slots : 0; }
// not part of the public header: 1 when the context's upload and download streams sit on two copy engines, 0 when they take turns, -1 before the first large host call
int zl_cctx_copy_duplex(const zl_cctx* c) { return c ? c->copy_pair_duplex : -1; }
size_t zl_cctx_far_ms(const zl_cctx* c, double* ms) {
    if (!c || !ms || !c->profile || !c->ev_valid) return ZL_ERR(ZL_E_STAGE_WRONG);
    *ms = c->far_ms;
    return 0;
}

#ifdef ZL_TRACE
// development build: the timeline the match kernel stamped (ZL_TRACE_BYTES / 8 words)
size_t zl_cctx_trace(const zl_cctx* c, unsigned long long* out, size_t words) {
    if (!c || !out || !c->d_stats.p) return ZL_ERR(ZL_E_STAGE_WRONG);
    if (words > ZL_TRACE_BYTES / 8) words = ZL_TRACE_BYTES / 8;
    (void)hipMemcpy(out, (const u8*)c->d_stats.p + 128, words * 8, hipMemcpyDeviceToHost);
    return words;
}
#endif
// not part of the public header: match-kernel phase cycle sums {find, parse, emit, insert, iterations}
size_t zl_cctx_phase_stats(const zl_cctx* c, unsigned long long* out8) {
    if (!c || !out8 || !c->profile) return ZL_ERR(ZL_E_STAGE_WRONG);
    for (int i = 0; i < 16; i++) out8[i] = c->stats[i];
    return 0;
}

}  // extern "C"

namespace {
// Host buffers, large input: the three legs of the call -- upload, kernels, download -- overlap.  The input goes up
// in sections of ZL_PIPE_CHUNKS chunks (64 MiB) on a copy stream; each section's kernels run on one of ZL_PIPE_SETS
// streams with its own scratch as soon as its bytes have arrived (a section alone fills a third of the device: three in
// flight keep it busy); a second host thread brings every section's frames down as soon as they are packed, while this
// thread is still uploading (PCIe carries both directions at once; a pageable hipMemcpy occupies its caller).
// The bytes written are those of the one-stream path: sections are the unit of far history in both (ZL_FAR_SECTION_LOG).
#define ZL_PIPE_CHUNKS ((size_t)1 << ZL_FAR_SECTION_LOG)
#define ZL_PIPE_SETS 3
#define ZL_PIPE_MIN_BYTES ((size_t)192 << 20)   // below this one launch group over everything is faster than sections
// A RANGE of an input: chunks [chunk0, chunk0 + nchunks) of an input of n_total bytes in total_chunks chunks (chunk0 a
// multiple of the section size) -- the whole input, or one device's shard of it (compress2_sharded).  src points at the
// range's first byte.  With dst: the range's frames are brought down there as they are packed (cap bytes) and the byte
// count is returned; without: they are left packed in c->d_dst and the count is returned.
// hs: ONE frame with a content checksum whose ranges arrive call by call (a pledged stream): the running XXH64.  The range's
// bytes are added to it here; the range that holds the frame's last chunk appends the digest.
size_t compress2_pipelined(zl_cctx* c, u8* dst, size_t cap, const u8* src, size_t n_total, bool one_frame, size_t chunk0 = 0, size_t nchunks_range = 0, size_t total_chunks = 0, zlx::Xxh64* hs = nullptr) {
    { const size_t pd = cctx_prepare_dict(c, cctx_stream(c, nullptr)); if (ZL_IS_ERR(pd)) return pd; }
    const bool whole = nchunks_range == 0;
    const size_t all_chunks = whole ? chunks_of(n_total) : total_chunks;
    const size_t nchunks = whole ? all_chunks : nchunks_range;
    const size_t range_off = chunk0 * (size_t)ZL_CHUNK;
    const size_t n = std::min<size_t>(nchunks * (size_t)ZL_CHUNK, n_total - range_off);  // bytes of the range
    {   // the range's chunk table: source offsets are positions in the INPUT (the far search and the frame header reason in those)
        if (!c->h_chunks.ensure(nchunks * sizeof(zl_chunk_desc))) return kGpuUnavailable;
        zl_chunk_desc* t = (zl_chunk_desc*)c->h_chunks.p;
        for (size_t i = 0; i < nchunks; i++) {
            t[i].src_off = range_off + (u64)i * ZL_CHUNK;
            t[i].src_size = (u32)std::min<size_t>(ZL_CHUNK, n - i * (size_t)ZL_CHUNK);
            t[i].flags = 0;
        }
        c->any_small = false;
        c->one_input = true;
        c->input_total = n_total;
    }
    const size_t ngroups = (nchunks + ZL_PIPE_CHUNKS - 1) / ZL_PIPE_CHUNKS;
    const size_t bound = zl_compressBound(n);
    const u8* const d_src_base = nullptr;  // (set below: the device buffer holds the range only; kernels index it by input position)
    (void)d_src_base;
    if (!c->d_src.ensure(n + 16) || !c->d_dst.ensure(bound + 16) || !c->d_chunks.ensure(nchunks * sizeof(zl_chunk_desc)) || !c->d_fsize.ensure(nchunks * sizeof(u32)) ||
        !c->d_foff.ensure(nchunks * sizeof(u64)) || !c->d_totals.ensure((ngroups + 1) * sizeof(u64)) || !c->h_totals.ensure((ngroups + 1) * sizeof(u64)))
        return kGpuUnavailable;
    {   // the upload and the download stream on two copy engines: the process's pair for this device (zl_copy_pair.h)
        const ZlCopyPair* pair = zl_copy_pair();
        if (!pair) return kGpuUnavailable;
        c->copy_stream = pair->up;
        c->down_stream = pair->down;
        c->copy_pair_duplex = pair->duplex;
    }
    for (int k = 0; k < ZL_PIPE_SETS; k++)
        if (!c->sets[k].st && hipStreamCreateWithFlags(&c->sets[k].st, hipStreamNonBlocking) != hipSuccess) return kGpuUnavailable;
    while (c->pipe_events.size() < 3 * ngroups + 1) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return kGpuUnavailable;
        c->pipe_events.push_back(e);
    }
    hipEvent_t* ev_up = c->pipe_events.data();             // section g has arrived
    hipEvent_t* ev_scan = ev_up + ngroups;                 // d_totals[g + 1] is valid
    hipEvent_t* ev_done = ev_scan + ngroups;               // section g's frames are packed in d_dst
    hipEvent_t ev_table = c->pipe_events[3 * ngroups];
    c->ev_valid = false;
    if (hipMemcpyAsync(c->d_chunks.p, c->h_chunks.p, nchunks * sizeof(zl_chunk_desc), hipMemcpyHostToDevice, c->copy_stream) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (hipMemsetAsync(c->d_totals.p, 0, sizeof(u64), c->copy_stream) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (hipEventRecord(ev_table, c->copy_stream) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    for (int k = 0; k < ZL_PIPE_SETS; k++) if (hipStreamWaitEvent(c->sets[k].st, ev_table, 0) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    int dev = 0;
    (void)hipGetDevice(&dev);
    // the download thread: section after section, as their frames become ready
    std::atomic<size_t> enqueued{0};
    std::atomic<bool> failed{false};
    size_t down_result = 0, down_total = 0;
    // the content checksum of ONE frame (zstd.c:27733): a third thread runs XXH64 over the range's bytes where they lie, in host
    // memory, beside the upload.  (A shard -- no dst -- leaves it to the caller, who sees the whole input.)
    const bool hashing = one_frame && c->checksum && dst;
    zlx::Xxh64 local_hash;
    zlx::Xxh64* const hash = hs ? hs : &local_hash;
    std::thread hasher;
    if (hashing) hasher = std::thread([&]() { hash->update(src, n); });
    std::thread down([&]() {
        (void)hipSetDevice(dev);
        u64 have = 0;
        for (size_t g = 0; g < ngroups && dst; g++) {
            while (enqueued.load(std::memory_order_acquire) <= g) { if (failed.load()) return; std::this_thread::yield(); }
            if (hipEventSynchronize(ev_done[g]) != hipSuccess) { down_result = ZL_ERR(ZL_E_GENERIC); return; }
            u64 upto = 0;
            if (hipMemcpyAsync(&upto, (const u64*)c->d_totals.p + g + 1, 8, hipMemcpyDeviceToHost, c->down_stream) != hipSuccess || hipStreamSynchronize(c->down_stream) != hipSuccess) { down_result = ZL_ERR(ZL_E_GENERIC); return; }
            if (upto > cap) { down_result = ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL); return; }
            if (upto > have && (hipMemcpyAsync(dst + have, (const u8*)c->d_dst.p + have, upto - have, hipMemcpyDeviceToHost, c->down_stream) != hipSuccess || hipStreamSynchronize(c->down_stream) != hipSuccess)) { down_result = ZL_ERR(ZL_E_GENERIC); return; }
            have = upto;
        }
        down_total = (size_t)have;
    });
    size_t err = 0;
#ifdef ZL_DEV_SWITCHES
    static const bool dbg_pipe = getenv("ZL_DEBUG_PIPE") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_start = now_ms();
#endif
    for (size_t g = 0; g < ngroups && !err; g++) {
        const size_t first = g * ZL_PIPE_CHUNKS;
        const u32 cnt = (u32)std::min<size_t>(ZL_PIPE_CHUNKS, nchunks - first);
        const size_t off = first * (size_t)ZL_CHUNK, len = std::min<size_t>((size_t)cnt * ZL_CHUNK, n - off);
        if (hipMemcpyAsync((u8*)c->d_src.p + off, src + off, len, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess || hipEventRecord(ev_up[g], c->copy_stream) != hipSuccess) { err = ZL_ERR(ZL_E_GENERIC); break; }
        zl_cctx_s::EncSet& es = c->sets[g % ZL_PIPE_SETS];
        if (hipStreamWaitEvent(es.st, ev_up[g], 0) != hipSuccess) { err = ZL_ERR(ZL_E_GENERIC); break; }
        err = cctx_enqueue_group(c, es, (u8*)c->d_dst.p, (const u8*)c->d_src.p - range_off, first, cnt, all_chunks, g, es.st, one_frame ? (u64)n_total : 0, false, g ? ev_scan[g - 1] : nullptr, ev_scan[g], chunk0);
        if (ZL_IS_ERR(err)) break;
        err = 0;
        if (hipEventRecord(ev_done[g], es.st) != hipSuccess) { err = ZL_ERR(ZL_E_GENERIC); break; }
        enqueued.store(g + 1, std::memory_order_release);
    }
    if (err) failed.store(true);
#ifdef ZL_DEV_SWITCHES
    const double t_enq = now_ms();
    if (dbg_pipe) { (void)hipStreamSynchronize(c->copy_stream); }
    const double t_up = now_ms();
#endif
    down.join();
#ifdef ZL_DEV_SWITCHES
    if (dbg_pipe) fprintf(stderr, "[zl pipe] enqueued (uploads handed over) %.2f ms, uploads done %.2f ms, downloads done %.2f ms\n", t_enq - t_start, t_up - t_start, now_ms() - t_start);
#endif
    if (hashing) hasher.join();
    for (int k = 0; k < ZL_PIPE_SETS; k++) (void)hipStreamSynchronize(c->sets[k].st);
    (void)hipStreamSynchronize(c->copy_stream);
    if (err) return err;
    if (down_result) return down_result;
    c->pending_chunks = nchunks;
    c->pending_groups = ngroups;
    if (!dst) {  // a shard: its frames stay on its device until every shard's size is known
        u64 upto = 0;
        if (hipMemcpy(&upto, (const u64*)c->d_totals.p + ngroups, 8, hipMemcpyDeviceToHost) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        return (size_t)upto;
    }
    const bool footer = hashing && chunk0 + nchunks == all_chunks;  // this range ends the frame
    if (footer) {
        down_total = append_checksum(dst, cap, down_total, hash->low32());
        if (ZL_IS_ERR(down_total)) return down_total;
    }
    if (wants_seek_table(c, n, one_frame) && nchunks > 1 && whole) {
        if (!c->h_fsize.ensure(nchunks * sizeof(u32)) || hipMemcpy(c->h_fsize.p, c->d_fsize.p, nchunks * sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (footer) ((u32*)c->h_fsize.p)[nchunks - 1] += 4;  // (the table's sizes add up to the bytes in front of it)
        return append_seek_table(c, dst, cap, down_total, nchunks, n, one_frame);
    }
    return down_total;
}

// Several devices behind one call (the reference is one process, src/init.c:90-99, so the sharding has to sit
// behind the ABI): the input's sections are dealt out in contiguous runs, one per device slot -- frames and sections are
// self-contained, so no device needs another's bytes --, every slot uploads and compresses its run on its own device
// from its own host thread (its own PCIe link: no root funnel), and once all sizes are known every slot downloads its
// frames to their final place.  The bytes written are those one device writes.  Slot s runs on device s mod the number
// of devices: ZL_c_devices above that number puts several slots on one device (which is how the path is tested on one).
size_t compress2_sharded(zl_cctx* c, u8* dst, size_t cap, const u8* src, size_t n, bool one_frame, int slots) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return kGpuUnavailable;
    const size_t nchunks = chunks_of(n), nsec = (nchunks + ZL_PIPE_CHUNKS - 1) / ZL_PIPE_CHUNKS;
    if ((size_t)slots > nsec) slots = (int)nsec;
    while (c->peers.size() < (size_t)slots) {
        zl_cctx* p = zl_createCCtx();
        if (!p) return ZL_ERR(ZL_E_MEMORY);
        c->peers.push_back(p);
    }
    std::vector<size_t> first(slots + 1), bytes(slots, 0);
    for (int s = 0; s <= slots; s++) first[s] = std::min(nchunks, (nsec * (size_t)s / (size_t)slots) * ZL_PIPE_CHUNKS);
    int home = 0;
    (void)hipGetDevice(&home);
    auto run = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int s = 0; s < slots; s++) th.emplace_back([&, s]() { (void)hipSetDevice(s % ndev); fn(s); });
        for (auto& t : th) t.join();
    };
    const bool hashing = one_frame && c->checksum;  // ONE frame's content checksum covers every slot's bytes: hashed here, beside the slots' work
    zlx::Xxh64 hash;
    std::thread hasher;
    if (hashing) hasher = std::thread([&]() { hash.update(src, n); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{hasher};
    run([&](int s) {
        zl_cctx* p = c->peers[s];
        p->level = c->level; p->checksum = c->checksum; p->layout = c->layout; p->dict = c->dict; p->deep_denied = c->deep_denied; p->fault = c->fault & 1;
        if (p->dict != c->dict || p->dict_id != c->dict_id) { p->dict_uploaded = false; p->dict_formatted = c->dict_formatted; p->dict_id = c->dict_id; }
        const size_t cnt = first[s + 1] - first[s];
        bytes[s] = cnt ? compress2_pipelined(p, nullptr, 0, src + first[s] * (size_t)ZL_CHUNK, n, one_frame, first[s], cnt, nchunks) : 0;
    });
    size_t total = 0;
    std::vector<size_t> at(slots);
    for (int s = 0; s < slots; s++) { if (ZL_IS_ERR(bytes[s])) return bytes[s]; at[s] = total; total += bytes[s]; }
    if (total > cap) return ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);
    std::vector<size_t> res(slots, 0);
    run([&](int s) {
        if (bytes[s] && hipMemcpy(dst + at[s], c->peers[s]->d_dst.p, bytes[s], hipMemcpyDeviceToHost) != hipSuccess) res[s] = ZL_ERR(ZL_E_GENERIC);
    });
    (void)hipSetDevice(home);
    for (int s = 0; s < slots; s++) if (res[s]) return res[s];
    if (hashing) {
        hasher.join();
        total = append_checksum(dst, cap, total, hash.low32());
        if (ZL_IS_ERR(total)) return total;
    }
    if (wants_seek_table(c, n, one_frame) && nchunks > 1) {  // every slot's per-chunk sizes, in order
        if (!c->h_fsize.ensure(nchunks * sizeof(u32))) return kGpuUnavailable;
        size_t err = 0;
        run([&](int s) {
            const size_t cnt = first[s + 1] - first[s];
            if (cnt && hipMemcpy((u32*)c->h_fsize.p + first[s], c->peers[s]->d_fsize.p, cnt * sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) err = ZL_ERR(ZL_E_GENERIC);
        });
        (void)hipSetDevice(home);
        if (err) return err;
        if (hashing) ((u32*)c->h_fsize.p)[nchunks - 1] += 4;
        return append_seek_table(c, dst, cap, total, nchunks, n, one_frame);
    }
    return total;
}
}  // namespace

extern "C" {

// replaces ZSTD_compress2 (zstd.h:599): host buffers in and out.
static size_t compress2_impl(zl_cctx* c, void* dst, size_t cap, const void* src, size_t n);
static size_t compress2_body(zl_cctx* c, void* dst, size_t cap, const void* src, size_t n);
size_t zl_compress2(zl_cctx* c, void* dst, size_t cap, const void* src, size_t n) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    if (!c->cs.started) c->deep_denied = false;  // (a one-shot call asks for the deep finder's memory afresh; a streaming session keeps what its first job got)
    return compress2_impl(c, dst, cap, src, n);
}
static size_t compress2_impl(zl_cctx* c, void* dst, size_t cap, const void* src, size_t n) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    return with_deep_fallback(c, [&]() { return compress2_body(c, dst, cap, src, n); });
}
static size_t compress2_body(zl_cctx* c, void* dst, size_t cap, const void* src, size_t n) {
    if (!device_ok()) return kGpuUnavailable;
    if (c->pending) return ZL_ERR(ZL_E_STAGE_WRONG);
    if (cap && !dst) return ZL_ERR(ZL_E_DSTBUFFER_NULL);
    if (n >= ZL_PIPE_MIN_BYTES && c->pipeline) {
        int slots = c->devices;
        if (slots == 0) { int nd = 1; (void)hipGetDeviceCount(&nd); slots = (int)std::min<size_t>((size_t)nd, n / ((size_t)512 << 20)); }  // auto: a device per 512 MiB at least
        c->last_slots = 1;
        if (slots > 1 && c->dict.empty()) { c->last_slots = (int)std::min<size_t>((size_t)slots, (chunks_of(n) + ZL_PIPE_CHUNKS - 1) / ZL_PIPE_CHUNKS); return compress2_sharded(c, (u8*)dst, cap, (const u8*)src, n, use_one_frame(c, n, true), slots); }
        return compress2_pipelined(c, (u8*)dst, cap, (const u8*)src, n, use_one_frame(c, n, true));
    }
    hipStream_t st = cctx_stream(c, nullptr);
    const size_t bound = zl_compressBound(n);
    if (!c->d_src.ensure(n + 16) || !c->d_dst.ensure(bound + 16)) return kGpuUnavailable;
    if (n && hipMemcpyAsync(c->d_src.p, src, n, hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    const size_t nchunks = fill_chunk_table(c, n);
    if (ZL_IS_ERR(nchunks)) return nchunks;
    const bool one = use_one_frame(c, n, true);
    const size_t e = cctx_enqueue(c, (u8*)c->d_dst.p, (const u8*)c->d_src.p, nchunks, st, one ? (u64)n : 0);
    if (ZL_IS_ERR(e)) return e;
    const bool hashing = one && c->checksum;  // ONE frame's content checksum: XXH64 on this thread while the kernels run
    zlx::Xxh64 hash;
    if (hashing) hash.update(src, n);
    if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    size_t total = (size_t)((const u64*)c->h_totals.p)[c->pending_groups];
    if (total > cap) return ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);
    if (hipMemcpyAsync(dst, c->d_dst.p, total, hipMemcpyDeviceToHost, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    if (hashing) {
        total = append_checksum((u8*)dst, cap, total, hash.low32());
        if (ZL_IS_ERR(total)) return total;
        ((u32*)c->h_fsize.p)[nchunks - 1] += 4;
    }
    if (wants_seek_table(c, n, one) && nchunks > 1) return append_seek_table(c, (u8*)dst, cap, total, nchunks, n, one);
    return total;
}

// n independent inputs in one launch: every input becomes its own frame (inputs above one
// chunk become several concatenated frames, exactly as zl_compress2 would emit them).
size_t zl_compress_batch(zl_cctx* c, size_t n, void* const* dsts, const size_t* caps, const void* const* srcs,
                         const size_t* sizes, size_t* results) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    if (!device_ok()) return kGpuUnavailable;
    if (c->pending) return ZL_ERR(ZL_E_STAGE_WRONG);
    if (n == 0) return 0;
    hipStream_t st = cctx_stream(c, nullptr);
    // (ZL_c_profile: the call's phases by the host clock, the stream drained between them -- a diagnostic, the drains cost a little)
    const bool prof = c->profile != 0;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double tp[6] = {now_ms(), 0, 0, 0, 0, 0};
    size_t nchunks = 0, in_total = 0, bound_total = 0;
    std::vector<size_t> first_chunk(n + 1), in_off(n);
    for (size_t i = 0; i < n; i++) {
        first_chunk[i] = nchunks;
        in_off[i] = in_total;
        nchunks += chunks_of(sizes[i]);
        in_total += (sizes[i] + 15) & ~(size_t)15;
        bound_total += zl_compressBound(sizes[i]);
    }
    first_chunk[n] = nchunks;
    if (!c->h_chunks.ensure(nchunks * sizeof(zl_chunk_desc)) || !c->h_stage_in.ensure(in_total + 16) || !c->h_stage_out.ensure(bound_total + 16) ||
        !c->d_src.ensure(in_total + 16) || !c->d_dst.ensure(bound_total + 16))
        return kGpuUnavailable;
    zl_chunk_desc* t = (zl_chunk_desc*)c->h_chunks.p;
    c->any_small = false;
    c->one_input = false;
    size_t largest = 0;
    for (size_t i = 0; i < n; i++) {
        largest = std::max(largest, sizes[i]);
        if (chunks_of(sizes[i]) == 1) c->any_small = true;
    }
    // Round 6: a large batch is packed in PIECES, each piece's upload issued as soon as it lies in the pinned buffer -- the pool packs the next piece
    // while the link carries this one (and, below, unpacks a piece while the next comes down): pack + upload of a 100 000-object call 1.4 -> ms.
    // (ZL_c_profile keeps one piece: its phases are measured with the stream drained between them.)
    const size_t pieces = (prof || n < 16384) ? 1 : ZL_BATCH_PIECES;
    // The copies ride the process's copy pair (zl_copy_pair.h), not the kernels' stream: a stream keeps the copy engine of its first copy, and
    // the context's own was seen to carry 30 MB up at 26 GB/s where the pair -- made with both directions checked for their rate -- carries 50.
    const ZlCopyPair* const pair = zl_copy_pair();
    const hipStream_t up = pair ? pair->up : st, down = pair ? pair->down : st;
    while (c->pipe_events.size() < pieces + 1) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return kGpuUnavailable;
        c->pipe_events.push_back(ev);
    }
    for (size_t pc = 0; pc < pieces; pc++) {
        const size_t b0 = n * pc / pieces, b1 = n * (pc + 1) / pieces;
        zl_parallel_for(b1 - b0, ZL_BATCH_GRAIN, [&](size_t b, size_t e) {
            for (size_t i = b0 + b; i < b0 + e; i++) {
                if (sizes[i]) memcpy((u8*)c->h_stage_in.p + in_off[i], srcs[i], sizes[i]);
                const size_t k = chunks_of(sizes[i]);
                for (size_t j = 0; j < k; j++) {
                    zl_chunk_desc& d = t[first_chunk[i] + j];
                    d.src_off = in_off[i] + j * (size_t)ZL_CHUNK;
                    d.src_size = (u32)std::min<size_t>(ZL_CHUNK, sizes[i] - j * (size_t)ZL_CHUNK);
                    d.flags = k == 1 ? 1u : 0u;
                }
            }
        });
        if (pc == 0) tp[1] = now_ms();
        const size_t o0 = in_off[b0], o1 = b1 < n ? in_off[b1] : in_total;
        if (o1 > o0 && hipMemcpyAsync((u8*)c->d_src.p + o0, (const u8*)c->h_stage_in.p + o0, o1 - o0, hipMemcpyHostToDevice, up) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    }
    if (up != st && (hipEventRecord(c->pipe_events[pieces], up) != hipSuccess || hipStreamWaitEvent(st, c->pipe_events[pieces], 0) != hipSuccess)) return ZL_ERR(ZL_E_GENERIC);
    if (prof) { (void)hipStreamSynchronize(st); }
    tp[2] = now_ms();
    c->chunk_cap = (u32)std::min<size_t>(largest, ZL_CHUNK);  // small objects: small scratch per object, one launch group (enc_strides)
    const size_t e = cctx_enqueue(c, (u8*)c->d_dst.p, (const u8*)c->d_src.p, nchunks, st);
    c->chunk_cap = ZL_CHUNK;
    if (ZL_IS_ERR(e)) return e;
    if (hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    tp[3] = now_ms();
    const size_t total = (size_t)((const u64*)c->h_totals.p)[c->pending_groups];
    const u32* fs = (const u32*)c->h_fsize.p;
    std::vector<size_t>& out_off = in_off;  // (reused: where each object's frames start in the packed output)
    size_t off = 0;
    for (size_t i = 0; i < n; i++) {
        size_t sz = 0;
        for (size_t j = first_chunk[i]; j < first_chunk[i + 1]; j++) sz += fs[j];
        out_off[i] = off;
        results[i] = sz;
        off += sz;
    }
    if (off != total) return ZL_ERR(ZL_E_GENERIC);
    // the frames come down piece by piece, a piece is unpacked while the next one travels (the kernels' stream has been drained above)
    for (size_t pc = 0; pc < pieces; pc++) {
        const size_t b0 = n * pc / pieces, b1 = n * (pc + 1) / pieces;
        const size_t o0 = out_off[b0], o1 = b1 < n ? out_off[b1] : total;
        if (o1 > o0 && hipMemcpyAsync((u8*)c->h_stage_out.p + o0, (const u8*)c->d_dst.p + o0, o1 - o0, hipMemcpyDeviceToHost, down) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (hipEventRecord(c->pipe_events[pc], down) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    }
    for (size_t pc = 0; pc < pieces; pc++) {
        const size_t b0 = n * pc / pieces, b1 = n * (pc + 1) / pieces;
        if (hipEventSynchronize(c->pipe_events[pc]) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (pc + 1 == pieces) tp[4] = now_ms();
        zl_parallel_for(b1 - b0, ZL_BATCH_GRAIN, [&](size_t b, size_t e) {
            for (size_t i = b0 + b; i < b0 + e; i++) {
                if (results[i] > caps[i]) results[i] = ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);
                else memcpy(dsts[i], (const u8*)c->h_stage_out.p + out_off[i], results[i]);
            }
        });
    }
    if (prof) { tp[5] = now_ms(); for (int k = 0; k < 5; k++) c->batch_ms[k] = tp[k + 1] - tp[k]; }
#ifdef ZL_DEV_SWITCHES
    else if (pieces > 1) { tp[5] = now_ms(); fprintf(stderr, "[zl c-batch] first piece packed %.3f | all issued %.3f | kernels done %.3f | last piece down %.3f | unpacked %.3f\n", tp[1] - tp[0], tp[2] - tp[0], tp[3] - tp[0], tp[4] - tp[0], tp[5] - tp[0]); }
#endif
    if (c->profile >= 2 && c->ev_valid && c->d_stats.p) (void)hipMemcpy(c->stats, c->d_stats.p, 128, hipMemcpyDeviceToHost);
    return 0;
}

// not part of the public header (diagnostics, ZL_c_profile): the phases of the last zl_compress_batch call in milliseconds:
// {pack into pinned memory, upload, kernels, download, unpack into the caller's buffers}
size_t zl_cctx_batch_ms(const zl_cctx* c, double* ms) {
    if (!c || !ms || !c->profile) return ZL_ERR(ZL_E_STAGE_WRONG);
    for (int k = 0; k < 5; k++) ms[k] = c->batch_ms[k];
    return 0;
}

// ---------------------------------------------------------------------------------------
// streaming: ZSTD_compressStream2 and the session calls (zstd.c:23793, :22775, :24835-24990), as a
// host-side job queue in front of zl_compress2
// ---------------------------------------------------------------------------------------
#define ZL_STREAM_JOB ((size_t)64 << 20)  // bytes of input per launch: 512 chunks -- a launch lasts about as long as its slowest chunk, so small jobs waste the device (zstd.c:36848: the multi-threaded streaming path cuts jobs too)

size_t zl_CStreamInSize(void) { return ZL_CHUNK; }
size_t zl_CStreamOutSize(void) { return zl_compressBound(ZL_CHUNK); }

size_t zl_CCtx_setPledgedSrcSize(zl_cctx* c, unsigned long long pledged) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    if (c->cs.started || c->cs.finishing) return ZL_ERR(ZL_E_STAGE_WRONG);  // zstd.c:23795: only between sessions
    c->cs.pledged = pledged;
    return 0;
}

size_t zl_CCtx_reset(zl_cctx* c, zl_ResetDirective reset) {
    if (!c) return ZL_ERR(ZL_E_GENERIC);
    cs_wait(c);
    if (reset == ZL_reset_session_only || reset == ZL_reset_session_and_parameters) {
        for (auto& j : c->cs.job) { j.in.clear(); j.out.clear(); j.ret = 0; }
        c->cs.job_head = c->cs.job_count = 0;
        c->cs.stage.clear();
        c->cs.out.clear();
        c->cs.out_pos = 0;
        c->cs.pledged = ZL_CONTENTSIZE_UNKNOWN;
        c->cs.consumed = 0;
        c->cs.started = c->cs.emitted = c->cs.finishing = c->cs.joined = c->cs.flush_told = false;
        c->deep_denied = false;
        c->jobs_started = 0;
    }
    if (reset == ZL_reset_parameters || reset == ZL_reset_session_and_parameters) {
        if (c->cs.started || c->cs.finishing) return ZL_ERR(ZL_E_STAGE_WRONG);  // zstd.c:22790
        c->level = 3;
        c->checksum = 0;
        c->nb_workers = 0;
        c->dict.clear();
        c->dict_uploaded = c->dict_formatted = false;
        c->dict_id = c->dict_content_off = c->dict_use = 0;
    }
    return 0;
}

size_t zl_compressStream2(zl_cctx* c, zl_outBuffer* output, zl_inBuffer* input, zl_EndDirective end_op) {
    if (!c || !output || !input) return ZL_ERR(ZL_E_GENERIC);
    if (output->pos > output->size) return ZL_ERR(ZL_E_DSTSIZE_TOO_SMALL);  // zstd.c:24848-24850
    if (input->pos > input->size) return ZL_ERR(ZL_E_SRCSIZE_WRONG);
    if ((unsigned)end_op > (unsigned)ZL_e_end) return ZL_ERR(ZL_E_PARAM_OUTOFBOUND);
    auto& S = c->cs;
    auto hand_out = [&]() {
        const size_t k = std::min(output->size - output->pos, S.out.size() - S.out_pos);
        zl_memcpy_mt((u8*)output->dst + output->pos, S.out.p + S.out_pos, k);
        output->pos += k;
        S.out_pos += k;
        if (S.out_pos == S.out.size()) { S.out.clear(); S.out_pos = 0; }
    };
    // what the OLDEST job that ran beside the caller wrote becomes what waits to be handed out (behind whatever still waits there)
    auto collect = [&]() -> size_t {
        if (S.job_count == 0) return 0;
        auto& J = S.job[S.job_head];
        if (J.th.joinable()) J.th.join();
        zl_cctx* const w = S.worker[S.job_head];
        if (w && w->deep_denied) c->deep_denied = true;  // (the denial lasts for the session: a frame's header names ONE window)
        S.job_head ^= 1;
        S.job_count--;
        const size_t r = J.ret;
        J.in.clear();
        if (ZL_IS_ERR(r)) { J.out.clear(); return r; }
        if (S.out.empty()) {
            J.out.n = r;
            std::swap(S.out.p, J.out.p); std::swap(S.out.n, J.out.n); std::swap(S.out.cap, J.out.cap);
            S.out_pos = 0;
        } else if (!S.out.append(J.out.p, r)) { J.out.clear(); return ZL_ERR(ZL_E_MEMORY); }
        J.out.clear();
        S.emitted = true;
        return 0;
    };
    // a FULL job: on a thread and a worker context of its own, so that the caller's next pieces are copied while it is uploaded, compressed
    // and brought down -- and the job behind it started
    auto start_job = [&]() -> size_t {
        const int limit = (S.joined && c->checksum) ? 1 : 2;
        if (S.job_count >= limit) { const size_t e = collect(); if (ZL_IS_ERR(e)) return e; }
        const int slot = (S.job_head + S.job_count) & 1;
        auto& J = S.job[slot];
        if (!S.worker[slot]) S.worker[slot] = zl_createCCtx();
        zl_cctx* const w = S.worker[slot];
        if (!w) return ZL_ERR(ZL_E_MEMORY);
        {   // the worker compresses as this context would
            w->level = c->level; w->checksum = c->checksum; w->layout = c->layout; w->pipeline = c->pipeline; w->devices = c->devices; w->seek_table = c->seek_table;
            w->nb_workers = c->nb_workers; w->fault = c->fault & 1; w->deep_denied = c->deep_denied;
            if (w->dict != c->dict || w->dict_id != c->dict_id || w->dict_formatted != c->dict_formatted) { w->dict = c->dict; w->dict_id = c->dict_id; w->dict_formatted = c->dict_formatted; w->dict_uploaded = false; }
        }
        std::swap(S.stage.p, J.in.p); std::swap(S.stage.n, J.in.n); std::swap(S.stage.cap, J.in.cap);
        S.stage.clear();
        const size_t n = J.in.size(), bound = zl_compressBound(n);
        J.out.clear();
        if (!J.out.reserve(bound)) return ZL_ERR(ZL_E_MEMORY);
        const u64 pledged = S.pledged, consumed_end = S.consumed;
        const bool joined = S.joined;
        if (joined && ((consumed_end - n) % ZL_STREAM_JOB != 0 || !(n == ZL_STREAM_JOB || consumed_end == pledged))) return ZL_ERR(ZL_E_STAGE_WRONG);
        int dev = 0;
        (void)hipGetDevice(&dev);
        auto* const hash = &S.hash;
        auto* const Jp = &J;
        auto body = [w, Jp, hash, n, bound, pledged, consumed_end, joined, dev]() {
            (void)hipSetDevice(dev);
            const zlx::Xxh64 hash0 = *hash;  // (a call made a second time hashes its bytes a second time: from the same start)
            Jp->ret = joined ? with_deep_fallback(w, [&]() { *hash = hash0; return compress2_pipelined(w, Jp->out.p, bound, Jp->in.p, (size_t)pledged, true, (size_t)((consumed_end - n) / ZL_CHUNK), chunks_of(n), chunks_of((size_t)pledged), hash); })
                             : compress2_impl(w, Jp->out.p, bound, Jp->in.p, n);
        };
        const int fail_from = c->fault >> 4;  // (ZL_c_faultInjection: from the k-th job of the session on, thread creation "fails")
        try {
            if (fail_from > 0 && ++c->jobs_started >= fail_from) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
            J.th = std::thread(body);
            S.job_count++;
        } catch (...) {  // no thread to be had: the job runs here -- behind every job that is still under way, whose bytes go out first (they used
                         // to be replaced by this job's)
            while (S.job_count) { const size_t e = collect(); if (ZL_IS_ERR(e)) return e; }
            body();
            if (w->deep_denied) c->deep_denied = true;
            const size_t r = J.ret;
            J.in.clear();
            if (ZL_IS_ERR(r)) { J.out.clear(); return r; }
            if (S.out.empty()) {
                J.out.n = r;
                std::swap(S.out.p, J.out.p); std::swap(S.out.n, J.out.n); std::swap(S.out.cap, J.out.cap);
                S.out_pos = 0;
            } else if (!S.out.append(J.out.p, r)) {
                J.out.clear();
                return ZL_ERR(ZL_E_MEMORY);
            }
            J.out.clear();
            S.emitted = true;
        }
        return 0;
    };
    auto run_job = [&]() -> size_t {  // the collected input -> frames, appended to what waits to be handed out
        const size_t n = S.stage.size(), bound = zl_compressBound(n), old = S.out.size();
        u8* at = S.out.grow(bound);
        if (!at) return ZL_ERR(ZL_E_MEMORY);
        // With the size pledged (what the reference's file writer does, src/raw-file-out.c:62) and the one-frame layout the
        // jobs are consecutive runs of ONE frame -- the bytes zl_compress2 writes for the whole input, so that a reader that
        // takes the first frame gets all of it.  Jobs are whole sections (64 MiB), the unit the far search never crosses.
        // Without a pledge every job is a frame of its own (the total is not known when the first header is written).
        // A joined session stays joined: its jobs are only ever cut at section boundaries and at the pledged end (a flush
        // elsewhere hands out what is complete, see below) -- a frame of its own in the middle of the unterminated one would
        // be an invalid stream.
        const u64 pledged = S.pledged;
        size_t r;
        if (S.joined) {
            if ((S.consumed - n) % ZL_STREAM_JOB != 0 || !(n == ZL_STREAM_JOB || S.consumed == pledged)) { S.out.n = old; return ZL_ERR(ZL_E_STAGE_WRONG); }
            const zlx::Xxh64 hash0 = S.hash;
            r = with_deep_fallback(c, [&]() { S.hash = hash0; return compress2_pipelined(c, at, bound, S.stage.p, (size_t)pledged, true, (size_t)((S.consumed - n) / ZL_CHUNK), chunks_of(n), chunks_of((size_t)pledged), &S.hash); });
        } else
            r = compress2_impl(c, at, bound, S.stage.p, n);
        S.out.n = old + (ZL_IS_ERR(r) ? 0 : r);
        if (ZL_IS_ERR(r)) return r;
        S.stage.clear();
        S.emitted = true;
        return 0;
    };
    hand_out();
    while (!S.finishing && input->pos < input->size && S.out.empty()) {
        const size_t take = std::min(input->size - input->pos, ZL_STREAM_JOB - S.stage.size());
        if (!S.stage.append((const u8*)input->src + input->pos, take)) return ZL_ERR(ZL_E_MEMORY);
        if (!S.started) S.hash.reset();
        if (!S.started)  // with the size pledged and more than one job of it, the jobs are runs of ONE frame
            S.joined = S.pledged != ZL_CONTENTSIZE_UNKNOWN && S.pledged > ZL_STREAM_JOB && use_one_frame(c, (size_t)S.pledged, true);
        input->pos += take;
        S.consumed += take;
        if (take) S.flush_told = false;
        S.started = true;
        if (S.pledged != ZL_CONTENTSIZE_UNKNOWN && S.consumed > S.pledged) return ZL_ERR(ZL_E_SRCSIZE_WRONG);  // zstd.c:24140: more than was pledged
        if (S.stage.size() == ZL_STREAM_JOB) {
            const size_t e = start_job();
            if (ZL_IS_ERR(e)) return e;
            hand_out();
        }
    }
    while (end_op != ZL_e_continue && input->pos == input->size && S.out.empty() && S.job_count) {  // the jobs that are still running come out first, oldest first
        const size_t e = collect();
        if (ZL_IS_ERR(e)) return e;
        hand_out();
    }
    if (end_op != ZL_e_continue && input->pos == input->size && S.out.empty() && !S.finishing) {
        if (end_op == ZL_e_end && S.pledged != ZL_CONTENTSIZE_UNKNOWN && S.pledged != S.consumed) return ZL_ERR(ZL_E_SRCSIZE_WRONG);  // zstd.c:24140
        // the rest of the input; at the end of a session that has written nothing, the frame of an empty input.
        // ZL_e_flush inside a joined session cuts no job: the frame's runs end at 64 MiB boundaries and at the pledged end only,
        // so a flush hands out the jobs that are complete and keeps the rest collected (include/zl_hip.h).
        const bool cut = !(S.joined && end_op == ZL_e_flush && S.consumed != S.pledged);
        if (cut && (!S.stage.empty() || (end_op == ZL_e_end && !S.emitted))) {
            const size_t e = run_job();
            if (ZL_IS_ERR(e)) return e;
        }
        if (end_op == ZL_e_end) S.finishing = true;
        hand_out();
    } else if (S.finishing) {
        hand_out();
    }
    if (S.finishing && S.out.empty()) {  // session complete: the next call starts a new one
        S.pledged = ZL_CONTENTSIZE_UNKNOWN;
        S.consumed = 0;
        S.started = S.emitted = S.finishing = S.joined = S.flush_told = false;
        c->deep_denied = false;
        c->jobs_started = 0;
        return 0;
    }
    // ZL_e_flush inside a one-frame (pledged) session cuts no run: the input still collected is part of the answer, so that a caller
    // who reads 0 as "everything consumed so far can be decoded from the output" (zstd.h:841-880) is never told so wrongly
    // -- ONCE per stretch of input: a caller that repeats the flush "until it returns 0" (the loop zstd.h documents) would spin for ever on
    // bytes that only more input or ZL_e_end can release, so the same flush repeated without new input reports only what waits to be handed out.
    if (end_op == ZL_e_flush && S.joined && !S.finishing && !S.stage.empty() && !S.flush_told) { S.flush_told = true; return S.out.size() - S.out_pos + S.stage.size(); }
    return S.out.size() - S.out_pos;
}

}  // extern "C"
