// zl_decode3.hip -- large-frame pipeline: ONE zstd frame of many blocks (what the reference's
// zstd_compress() writes for any input above 128 KiB: src/raw-file.c:74 ZSTD_compress2 -> one frame,
// 128 KiB blocks, one window) decoded by the whole device.
//
// The reference walks such a frame block after block (zstd.c:41553 ZSTD_decompressFrame): block
// headers chain through their sizes, treeless literals and Repeat_Mode tables chain through earlier
// blocks, the repcode history chains through every sequence and matches reach back across blocks up
// to the window.  Here each chain is cut where it is cheap to cut:
//
//   zl3_walk_kernel     one lane follows the block headers (3 bytes per block)           zstd.c:43075
//                       -- or the host's own walk is uploaded (host entries), or a lane per chunk walks the chunk's few
//                       headers (frames of this library's one-frame layout, with the caller's chunk offsets)
//   zl3_parse_kernel    lane per block: literal / sequence section headers               zstd.c:43146, :43707
//   zl3_segment_kernel, zl3_plan_{a,b,c}_kernel   the frame becomes SEGMENTS: a block that describes tables plus the (up to 7) blocks behind it
//                       that only repeat them; for each segment, the earlier blocks whose
//                       Huffman tree / FSE tables it inherits are found with a running maximum
//   zl2_lit_kernel, zl2_seq_kernel (zl_decode2.hip)  the segments go through the lane pipeline's entropy
//                       kernels as if they were frames; inherited tables are rebuilt in extra rounds;
//                       repcodes stay symbolic for every block but the first
//   zl3_offsets_kernel  sixteen waves: output position and starting repcode history of every block
//   zl2_exec_kernel<true, false>  workgroup per run of 8 segments, all runs at once: bytes whose source lies in
//                       another segment are left as links (frame positions)
//   zl3_resolve_kernel  the deferred bytes: rounds in which every waiting byte walks up to 8 links towards the front
//   zl3_finish_kernel   content size, XXH64 (zstd.c:41646-41660), the frame's result word
//
// All integer work; output must be bit-exact with the reference decoder.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>

#include "zl_blocks.h"
#include "zl_decode3.h"
#include "zl_device.h"
#include "zl_format.h"
#include "zl_host.h"
#include "zl_kernels.h"
#include "zl_xxh64.h"

#ifdef ZL_DEV_SWITCHES
#define ZL_OOM_HERE (fprintf(stderr, "[zl oom] %s:%d\n", __FILE__, __LINE__), kGpuUnavailable)
#else
#define ZL_OOM_HERE kGpuUnavailable
#endif
namespace {

struct zl3_dev {  // device pointers of one large-frame decode
    zl3_info* info;
    zl3_hblk* hblk;        // per block: where its content starts and its 3-byte header (device walk, chunk-hinted walk, or uploaded from the host's own walk)
    zl2_blk_rec* braw;     // per block, in frame order: the parsed record
    u32* seg_first;        // per segment: its first block (nsegs + 1 entries)
    u32* blk_slot;         // per block: index of its record in brec = segment * ZL2_MAXB + position inside the segment
    zl2_blk_rec* brec;     // per segment ZL2_MAXB records: what the lane pipeline's kernels read
    zl2_frame_rec* frec;
    zl2_frame_plan* plan;
    zl_frame_desc* frames;
    zl2_blk_rec* ext;
    u32* blk_dst;
    u32* rep_start;
    u32* seg_tmp;   // per segment 16 words (zl3_plan_kernel): [0..5] its own numbers, [6..13] the scanned ones
    u32* max_blocks;
    u32 cap_blocks;
    u32 dict_id;        // id of the dictionary loaded into the context (0: none, or raw content)
    u32 dict_tables;    // the context holds a formatted dictionary: blocks may ride on ITS entropy tables ...
    const zl2_blk_rec* dict_ext;  // ... which the segments inherit from these records (zl3_hint::dict_ext), or null: such frames go to the general kernel
    u32 have_dict;      // the context holds a dictionary of any kind
};

__device__ __forceinline__ void info_err(zl3_info* info, u32 code) { atomicCAS(&info->err, 0u, code); }

// frame header (zstd.c:41050-41148) -> info; returns the header size or 0 after recording an error
__device__ __forceinline__ u32 zl3_frame_header(zl3_info* info, const zl_frame_desc& fd, const u8* src, bool whole = true, u32 ctx_dict_id = 0) {
    const u64 n = fd.src_size;
    if (n < 5) { info->err = ZL_E_SRCSIZE_WRONG; return 0; }
    if (zl_rd_le(src, 4) != ZL_MAGIC) { info->err = ZL_E_PREFIX_UNKNOWN; return 0; }
    const u32 fhd = src[4];
    const u32 dflag = fhd & 3, single = (fhd >> 5) & 1, fflag = fhd >> 6;
    if (fhd & 8) { info->err = ZL_E_FRAMEPARAM_UNSUPPORTED; return 0; }
    const u32 dbytes = dflag == 3 ? 4 : dflag;
    const u32 fbytes = fflag == 0 ? (single ? 1 : 0) : (1u << fflag);
    const u32 hs = 5 + (single ? 0 : 1) + dbytes + fbytes;
    if (n < hs) { info->err = ZL_E_SRCSIZE_WRONG; return 0; }
    u32 pos = 5;
    if (!single) {
        const u32 wlog = (src[pos++] >> 3) + 10;
        if (wlog > 31) { info->err = ZL_E_WINDOW_TOO_LARGE; return 0; }
    }
    const u32 did = zl_rd_le(src + pos, dbytes);
    pos += dbytes;
    u64 fcs = ZL_CONTENTSIZE_UNKNOWN;
    if (fflag == 0) { if (single) fcs = src[pos]; }
    else if (fflag == 1) fcs = (u64)zl_rd_le(src + pos, 2) + 256;
    else if (fflag == 2) fcs = zl_rd_le(src + pos, 4);
    else fcs = (u64)zl_rd_le(src + pos, 4) | ((u64)zl_rd_le(src + pos + 4, 4) << 32);
    if (did && did != ctx_dict_id) { info->err = ZL_E_DICT_WRONG; return 0; }  // zstd.c:41318: the frame names a dictionary the context does not hold
    // (a frame that does not record its size -- what ZSTD_compressStream2 writes when nothing is pledged -- is
    // bounded by its block count: the host sizes the links from that)
    if (!whole) { info->fcs = fd.dst_cap; info->has_checksum = 0; return hs; }  // the first run of a frame dealt out over several devices: it regenerates what its share of the output holds
    if (fcs != ZL_CONTENTSIZE_UNKNOWN && fcs > fd.dst_cap) { info->err = ZL_E_DSTSIZE_TOO_SMALL; return 0; }
    info->fcs = fcs;
    info->has_checksum = (fhd >> 2) & 1;
    return hs;
}

// ---------------------------------------------------------------------------------------
// block walk, three ways.  (1) one lane follows the 3-byte headers (zstd.c:43075): the only way when
// nothing else is known -- 0.7 us of DRAM latency per block.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void zl3_walk_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base) {
    if (threadIdx.x != 0) return;
    zl3_info* info = d.info;
    const u8* src = src_base + fd.src_off;
    const u64 n = fd.src_size;
    u32 pos = zl3_frame_header(info, fd, src, true, d.dict_id);
    if (!pos) return;
    u32 nb = 0;
    for (;;) {
        if (n - pos < 3) { info->err = ZL_E_SRCSIZE_WRONG; return; }
        const u32 bh = zl_rd_le(src + pos, 3);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
        pos += 3;
        if (type == 3) { info->err = ZL_E_CORRUPTION; return; }
        const u32 csize = (type == 1) ? 1 : bsize;
        if (n - pos < csize) { info->err = ZL_E_SRCSIZE_WRONG; return; }
        if (nb >= d.cap_blocks) { info->err = ZL_E_FALLBACK; return; }  // more, smaller blocks than planned for: the general kernel takes the frame
        if (type != 2 && bsize > ZL_BLOCK_MAX) { info->err = ZL_E_CORRUPTION; return; }  // zstd.c:43085 (block size above the maximum)
        d.hblk[nb].src_off = pos;
        d.hblk[nb].hdr = bh;
        nb++;
        pos += csize;
        if (last) break;
    }
    if (info->has_checksum && n - pos < 4) { info->err = ZL_E_CHECKSUM_WRONG; return; }
    info->nblocks = nb;
    info->end_pos = pos;
}

// (1b) the same walk, SPECULATIVELY in parallel (round 3; one lane took 5.0 ms for the 8192 blocks of a 1 GiB reference frame,
// 46 ms for the 65 536 of this library's own).  The frame is cut into parts of ZL3_SPEC_PART bytes.  The first block header inside
// a part lies within its first 128 KiB + 3 bytes (the block that straddles the part's start is at most that long), so every
// byte offset there is a CANDIDATE: a lane per candidate follows headers from it until the chain leaves the part, and dies at the
// first header that cannot be one (reserved type, a size above the block maximum, past the end of the frame) -- garbage
// survives one step in twenty.  The few survivors per part (the true chain, and whatever happened to fall onto it: chains
// self-synchronise) are listed with where they started, where they left the part and how many blocks they saw.  The host then
// links the parts -- part 0's chain starts behind the frame header, the next part's true chain starts where this one's left --
// and a lane per part walks its true chain once more to fill the block table.  Anything that does not link up (a damaged
// frame, a part with more survivors than its list holds) goes to the one-lane walk, which also names the error.
#define ZL3_SPEC_PART (2u << 20)
#define ZL3_SPEC_CANDS (ZL_BLOCK_MAX + 3u)
#define ZL3_SPEC_LIST 16u
struct zl3_spec_rec { u32 start, exit, count, last; };
// follows block headers from `pos` until the chain leaves [.., part_end) or ends the frame; false: not a chain
__device__ __forceinline__ bool zl3_spec_follow(const u8* src, u64 n, u64 pos, u64 part_end, zl3_spec_rec& r, zl3_hblk* out /* or null */, u32 out_cap) {
    u32 count = 0;
    r.start = (u32)pos; r.last = 0;
    for (;;) {
        if (n - pos < 3) return false;
        const u32 bh = zl_rd_le(src + pos, 3);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
        if (type == 3 || bsize > ZL_BLOCK_MAX) return false;  // (a compressed block above the maximum is an error of the frame: the one-lane walk reports it)
        pos += 3;
        const u32 csize = (type == 1) ? 1 : bsize;
        if (n - pos < csize) return false;
        if (out) { if (count >= out_cap) return false; out[count].src_off = (u32)pos; out[count].hdr = bh; }
        count++;
        pos += csize;
        if (last) { r.last = 1; break; }
        if (pos >= part_end) break;
    }
    r.exit = (u32)pos; r.count = count;
    return true;
}
__global__ __launch_bounds__(256) void zl3_spec_walk_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base, u32 nparts, u32* counts, zl3_spec_rec* lists) {
    const u32 k = blockIdx.y, cand = blockIdx.x * 256 + threadIdx.x;
    const u8* src = src_base + fd.src_off;
    const u64 n = fd.src_size;
    const u64 part_start = (u64)k * ZL3_SPEC_PART, part_end = part_start + ZL3_SPEC_PART;
    u64 pos;
    if (k == 0) {  // the one candidate of part 0: behind the frame header
        if (cand != 0) return;
        const u32 hs = zl3_frame_header(d.info, fd, src, true, d.dict_id);
        if (!hs) return;
        d.info->hdr_size = hs;
        pos = hs;
    } else {
        if (cand >= ZL3_SPEC_CANDS) return;
        pos = part_start + cand;
    }
    zl3_spec_rec r;
    if (pos >= n || !zl3_spec_follow(src, n, pos, part_end, r, nullptr, 0)) return;
    if (!r.last && r.exit >= part_end + ZL3_SPEC_CANDS) return;  // (cannot happen: a block is shorter than that)
    // a chain that meets a Last_Block must end where the frame does (half of all headers that garbage spells carry the bit:
    // thousands of "frames" of one block per part otherwise); a frame table with slack behind the frame goes to the one-lane walk
    if (r.last && (u64)r.exit + (((src[4] >> 2) & 1) ? 4u : 0u) != n) return;
    const u32 slot = atomicAdd(&counts[k], 1u);
    if (slot < ZL3_SPEC_LIST) lists[(size_t)k * ZL3_SPEC_LIST + slot] = r;
}
// lane per part: the part's true chain (start[k]; ZL3_NONE: no block starts in this part) into the block table
__global__ __launch_bounds__(64) void zl3_spec_fill_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base, u32 nparts, const u32* start, const u32* first, u32 nblocks, u32 end_pos) {
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    zl3_info* info = d.info;
    if (k == 0) {
        if (nblocks > d.cap_blocks) info_err(info, ZL_E_FALLBACK);  // more, smaller blocks than planned for: the general kernel takes the frame
        else if (info->has_checksum && fd.src_size - end_pos < 4) info_err(info, ZL_E_CHECKSUM_WRONG);
        info->nblocks = nblocks;
        info->end_pos = end_pos;
    }
    if (k >= nparts || start[k] == 0xFFFFFFFFu || nblocks > d.cap_blocks) return;
    zl3_spec_rec r;
    const u64 part_end = ((u64)k + 1) * ZL3_SPEC_PART;
    if (!zl3_spec_follow(src_base + fd.src_off, fd.src_size, start[k], part_end, r, d.hblk + first[k], d.cap_blocks - first[k])) info_err(info, ZL_E_CORRUPTION);  // (the same bytes gave a chain a moment ago)
}

// (2) the frame was written by this library's one-frame layout and the caller knows where every chunk's run of blocks
// starts (zl_decompress_device_hint): a lane per chunk follows the few headers of its chunk.  Two passes around a scan
// of the block counts.  The hint is only a hint: every chunk must end exactly where the next one starts, the frame's
// last block must be the last chunk's last block and no other -- otherwise ZL_E_FALLBACK_WALK sends the frame to (1).
#define ZL3_CHUNK_BLOCKS_MAX 64u
// (the last chunk's listed size covers the frame's footer too -- 4 bytes of content checksum when the header says so)
__device__ __forceinline__ u32 zl3_tail(const u8* src, u32 tail, u32 headerless) { return tail != ZL3_TAIL_FROM_HEADER ? tail : ((!headerless && ((src[4] >> 2) & 1)) ? 4u : 0u); }
__global__ __launch_bounds__(256) void zl3_chunk_walk_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base, const u64* chunk_off, u32 nchunks, u32* cnt, const u32* first, u32 pass, u32 headerless, u32 not_last, u32 tail) {
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    zl3_info* info = d.info;
    if (c >= nchunks) return;
    // pass 1 stores through first[], which the scan kernel only fills when pass 0 found every chunk sound AND the blocks fit the
    // table (cap_blocks): after any error there is nothing to store, and what first[] holds is stale
    if (pass == 1 && info->err) return;
    const u8* src = src_base + fd.src_off;
    u64 pos = chunk_off[c];
    u64 end = chunk_off[c + 1];
    if (c + 1 == nchunks) { const u32 tl = fd.src_size >= 5 ? zl3_tail(src, tail, headerless) : 0u; end = end >= tl ? end - tl : 0; }
    if (c == 0) {
        if (pass == 0) {
            if (headerless) { info->fcs = fd.dst_cap; info->has_checksum = 0; info->hdr_size = 0; }
            else { const u32 hs = zl3_frame_header(info, fd, src, !not_last, d.dict_id); if (!hs) return; info->hdr_size = hs; }
        }
        pos = info->hdr_size;
    }
    u32 nb = 0;
    bool bad = end > fd.src_size || pos > end;
    while (!bad && pos < end) {
        if (end - pos < 3 || nb >= ZL3_CHUNK_BLOCKS_MAX) { bad = true; break; }
        const u32 bh = zl_rd_le(src + pos, 3);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
        pos += 3;
        const u32 csize = (type == 1) ? 1 : bsize;
        if (type == 3 || end - pos < csize || (type != 2 && bsize > ZL_BLOCK_MAX)) { bad = true; break; }
        if (pass == 1) {
            const u32 slot = first[c] + nb;
            if (slot >= d.cap_blocks) { bad = true; break; }  // (cannot happen behind a sound scan; a guard for the table's bounds)
            d.hblk[slot].src_off = (u32)pos; d.hblk[slot].hdr = bh;
        }
        nb++;
        pos += csize;
        if (last != ((!not_last && c + 1 == nchunks && pos == end) ? 1u : 0u)) { bad = true; break; }  // Last_Block: the last chunk's last block, and only that one
    }
    if (bad || nb == 0) { info_err(info, ZL_E_FALLBACK_WALK); return; }
    if (pass == 0) cnt[c] = nb;
}
__global__ __launch_bounds__(1024) void zl3_chunk_scan_kernel(zl3_dev d, zl_frame_desc fd, const u32* cnt, u32* first, u32 nchunks, const u64* chunk_off, const u8* src_base, u32 headerless, u32 tail) {
    __shared__ u32 part[1024];
    zl3_info* info = d.info;
    if (info->err) return;
    const u32 t = threadIdx.x, per = (nchunks + 1023) / 1024;
    const u32 b = t * per, e = b + per < nchunks ? b + per : nchunks;
    u32 sum = 0;
    for (u32 i = b; i < e; i++) sum += cnt[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        u32 run = 0;
        for (u32 i = 0; i < 1024; i++) { const u32 v = part[i]; part[i] = run; run += v; }
        if (run > d.cap_blocks) info->err = ZL_E_FALLBACK;
        info->nblocks = run;
        const u32 tl = fd.src_size >= 5 ? zl3_tail(src_base + fd.src_off, tail, headerless) : 0u;
        const u64 end = chunk_off[nchunks] >= tl ? chunk_off[nchunks] - tl : 0;
        info->end_pos = (u32)end;
        if (info->has_checksum && fd.src_size - end < 4) info->err = ZL_E_CHECKSUM_WRONG;
    }
    __syncthreads();
    u32 run = part[t];
    for (u32 i = b; i < e; i++) { first[i] = run; run += cnt[i]; }
}

// (3) the host has walked the frame already (zl_findFrameCompressedSize does, src/raw-file.c:150) and uploads
// what it saw: nothing to do on the device but the checks the walk would have made.
__global__ __launch_bounds__(64) void zl3_host_walk_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base, u32 nblocks, u32 end_pos, u32 headerless, u32 size_unknown, u32 ndonors, uint4 donor) {
    if (threadIdx.x != 0) return;
    zl3_info* info = d.info;
    if (headerless) { info->fcs = size_unknown ? ZL_CONTENTSIZE_UNKNOWN : fd.dst_cap; info->has_checksum = 0; info->hdr_size = 0; }  // (a run of a frame's blocks: the caller holds the header and the footer)
    else if (!zl3_frame_header(info, fd, src_base + fd.src_off, true, d.dict_id)) return;
    if (nblocks + ndonors > d.cap_blocks) { info->err = ZL_E_FALLBACK; return; }
    if (info->has_checksum && fd.src_size - end_pos < 4) { info->err = ZL_E_CHECKSUM_WRONG; return; }
    info->nblocks = nblocks;
    info->end_pos = end_pos;
    info->ndonors = ndonors;
    const u32 dn[4] = {donor.x, donor.y, donor.z, donor.w};
    for (int q = 0; q < 4; q++) info->donor[q] = (ndonors && dn[q] < ndonors) ? nblocks + dn[q] : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------
// lane per block: section headers; whether the block opens a segment
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zl3_parse_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base) {
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    zl3_info* info = d.info;
    const u32 nall = info->nblocks + info->ndonors;  // (the donors of a run of blocks are parsed like its own blocks)
    if (info->err || (b & ~63u) >= nall) return;  // (whole waves stay: the count at the end is a ballot)
    if (b >= nall) { (void)__ballot(false); return; }
    const zl3_hblk hb = d.hblk[b];
    bool dependent = false;
    zl2_blk_rec r;
    r.type = (hb.hdr >> 1) & 3; r.size = hb.hdr >> 3; r.src_off = hb.src_off;
    r.lit_mode = 0; r.lit_size = 0; r.lit_a = 0; r.lit_b = 0; r.seq_off = 0; r.seq_len = 0;
    r.lit_dst = 0; r.seq_dst = 0; r.nseq = 0; r.streams = 0; r.seq_modes = 0; r.pad[0] = 0; r.pad[1] = 0;
    if (r.type == 2) {
        u32 lit_bytes = 0;
        const u32 e = parse_block_sections(src_base + fd.src_off + r.src_off, r.size, r, lit_bytes);
        if (e) { info_err(info, e); (void)__ballot(false); return; }
        r.pad[0] = lit_bytes;  // (scratch the block's Huffman literals need; consumed by zl3_plan_kernel)
        // A block that describes a Huffman tree or any sequence table opens a segment: the blocks behind it that only
        // repeat (treeless literals, Repeat_Mode) ride on its tables in the lane pipeline's kernels, one table build
        // for up to ZL2_MAXB blocks.  Reference-made blocks describe their own tables: one block per segment.
        // Reference-made blocks describe their own tables: one block per segment.  A block that needs no table at all
        // (raw literals, no sequences or predefined tables) rides on nothing: a segment of its own.
        const u32 m = r.seq_modes;
        const bool has_modes = r.nseq > 0 && m != 0x100u;
        const bool seq_repeats = has_modes && ((m >> 6) == 3 || ((m >> 4) & 3) == 3 || ((m >> 2) & 3) == 3);
        const bool seq_tables = has_modes && ((m >> 6) != 3 || ((m >> 4) & 3) != 3 || ((m >> 2) & 3) != 3);
        const bool rides = r.lit_mode == 3 || seq_repeats;           // uses tables of an earlier block
        r.pad[1] = (r.lit_mode == 2 || seq_tables || !rides) ? 1u : 0u;
        dependent = rides;
    } else r.pad[1] = 1;  // a raw or RLE block needs no entropy work
    if (b == 0) r.pad[1] = 1;
    d.braw[b] = r;
    // how many blocks ride on earlier tables decides whether segments are formed at all (zl3_segment_kernel)
    const u64 md = __ballot(b < info->nblocks && dependent && r.pad[1] == 0);  // (blocks that ride and bring nothing of their own: the ones a segment would spare a table rebuild)
    if ((threadIdx.x & 63) == 0 && md) atomicAdd(&info->ndependent, (u32)__popcll(md));
}

// ---------------------------------------------------------------------------------------
// segments: a segment starts at every block that opens one and after every ZL2_MAXB blocks of a run.  One
// workgroup, two scans over the blocks: the position of the last opener (running maximum), then the segment count.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wg_scan_incl(u32* buf /* 2 x 1024 */, u32 v, bool is_max) {
    const u32 t = threadIdx.x;
    u32* cur = buf;
    u32* nxt = buf + 1024;
    cur[t] = v;
    __syncthreads();
    for (u32 d = 1; d < 1024; d <<= 1) {
        u32 x = cur[t];
        if (t >= d) { const u32 y = cur[t - d]; x = is_max ? (y > x ? y : x) : x + y; }
        nxt[t] = x;
        __syncthreads();
        u32* tmp = cur; cur = nxt; nxt = tmp;
    }
    const u32 out = cur[t];
    __syncthreads();
    return out;
}
__global__ __launch_bounds__(1024) void zl3_segment_kernel(zl3_dev d) {
    __shared__ u32 buf[2048];
    __shared__ u32 carry[2];
    zl3_info* info = d.info;
    if (info->err) return;
    const u32 t = threadIdx.x, nb = info->nblocks;
    // Segments of several blocks take the lane pipeline's multi-block kernels, which a frame of single-table
    // blocks does not want (its blocks decode fastest on the fused single-block kernel; the few that ride on
    // earlier tables rebuild them from zl3_plan_kernel's inherited records).  Form segments only where riding
    // is the rule: this library's own frames (7 of 8 blocks ride), not the reference's (a few per thousand).
    const bool grouping = info->ndependent * 4u >= nb;
    if (t == 0) { carry[0] = 0; carry[1] = 0; }  // [0] last opener + 1 so far, [1] segments so far
    __syncthreads();
    for (u32 b0 = 0; b0 < nb; b0 += 1024) {
        const u32 b = b0 + t;
        const bool on = b < nb;
        const u32 opener = (on && (!grouping || d.braw[b].pad[1])) ? b + 1 : 0u;
        u32 last = wg_scan_incl(buf, opener, true);
        const u32 c0 = carry[0], c1 = carry[1];
        if (c0 > last) last = c0;  // (block 0 is an opener, so last >= 1 for every block)
        const u32 idx = on ? b - (last - 1) : 0u;
        const u32 start = (on && idx % ZL2_MAXB == 0) ? 1u : 0u;
        const u32 seg_incl = wg_scan_incl(buf, start, false);
        if (on) {
            const u32 seg = c1 + seg_incl - 1;
            d.blk_slot[b] = seg * ZL2_MAXB + idx % ZL2_MAXB;
            if (start) d.seg_first[seg] = b;
        }
        __syncthreads();
        if (t == 1023) { carry[0] = last; carry[1] = c1 + seg_incl; }
        __syncthreads();
    }
    if (t == 0) { info->nsegs = carry[1]; d.seg_first[carry[1]] = nb; *d.max_blocks = 1; }  // (max_blocks: raised by zl3_plan_a_kernel)
}

// ---------------------------------------------------------------------------------------
// segments: scratch offsets and inherited tables.  One workgroup.
// ---------------------------------------------------------------------------------------
#define ZL3_NONE 0xFFFFFFFFu
// the latest defining block: what the segments so far had (a), unless this segment has one of its own (b) -- which is later in the
// frame than anything in front of it, the donors of a run of blocks (whose indices lie BEHIND the run's own) included
__device__ __forceinline__ u32 max_def(u32 a, u32 b) { return (b == ZL3_NONE) ? a : b; }

// (round 4: three launches -- A and C over all segments on the whole device, B one workgroup's scan -- where one workgroup did all
// three: 0.6 ms for the 8192 segments of this library's 1 GiB frame, a CU copying 65 536 block records)
__global__ __launch_bounds__(256) void zl3_plan_a_kernel(zl3_dev d) {
    zl3_info* info = d.info;
    if (info->err) return;
    const u32 ns = info->nsegs;
    // ---- A: inside each segment: where its blocks' literals and sequences go, totals, last defining blocks
    {
        const u32 s = blockIdx.x * 256 + threadIdx.x;
        if (s >= ns) return;
        u32 lit = 0, seq = 0, def[4] = {ZL3_NONE, ZL3_NONE, ZL3_NONE, ZL3_NONE};
        const u32 b0 = d.seg_first[s], b1 = d.seg_first[s + 1];
        for (u32 b = b0; b < b1; b++) {
            zl2_blk_rec r = d.braw[b];
            if (r.type == 2) {
                r.lit_dst = lit;
                r.seq_dst = seq;
                lit += r.pad[0];
                seq += r.nseq + 2;  // + the totals and final-repcode records the sequence lanes append
                if (r.lit_mode == 2) def[0] = b;
                if (r.nseq > 0 && r.seq_modes != 0x100u) {
                    const u32 m = r.seq_modes;
                    if ((m >> 6) != 3) def[1] = b;
                    if (((m >> 4) & 3) != 3) def[2] = b;
                    if (((m >> 2) & 3) != 3) def[3] = b;
                }
            }
            d.brec[(size_t)s * ZL2_MAXB + (b - b0)] = r;
        }
        if (b1 - b0 > 1) atomicMax(d.max_blocks, b1 - b0);  // (which instantiation of the literal / sequence kernels the segments take: 1 = the single-block ones; set to 1 by zl3_segment_kernel)
        u32* o = d.seg_tmp + (size_t)s * 16;
        o[0] = (lit + 15) & ~15u; o[1] = seq; o[2] = def[0]; o[3] = def[1]; o[4] = def[2]; o[5] = def[3];
    }
}

__global__ __launch_bounds__(1024) void zl3_plan_b_kernel(zl3_dev d) {
    __shared__ u64 carry_sum[2];
    __shared__ u32 carry_def[4];
    zl3_info* info = d.info;
    if (info->err) return;
    const u32 t = threadIdx.x, ns = info->nsegs;
    if (t == 0) { carry_sum[0] = 0; carry_sum[1] = 0; for (int q = 0; q < 4; q++) carry_def[q] = info->ndonors ? info->donor[q] : ZL3_NONE; }
    __syncthreads();
    // ---- B: across segments: exclusive sums of the scratch needs, exclusive running maximum of the defining blocks
    // (round 4: a scan by all 1024 threads -- wave scans, the sixteen wave totals folded through LDS.  One lane per column walked
    // a tile's 1024 entries one after the other: 0.6 of the kernel's 0.87 ms on the 8192 segments of this library's 1 GiB frame)
    __shared__ u64 wsum[2][16];
    __shared__ u32 wdef[4][16];
    const u32 lane = t & 63, wave = t >> 6;
    for (u32 s0 = 0; s0 < ns; s0 += 1024) {
        const u32 cnt = ns - s0 < 1024 ? ns - s0 : 1024;
        const bool on = t < cnt;
        const u32* in = d.seg_tmp + (size_t)(s0 + t) * 16;
        u64 a[2] = {on ? (u64)in[0] : 0ull, on ? (u64)in[1] : 0ull};
        u32 m[4] = {on ? in[2] : ZL3_NONE, on ? in[3] : ZL3_NONE, on ? in[4] : ZL3_NONE, on ? in[5] : ZL3_NONE};
        const u64 own[2] = {a[0], a[1]};
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {  // inclusive scans inside the wave
            u64 x[2];
            u32 y[4];
            for (int q = 0; q < 2; q++) x[q] = ((u64)__shfl_up((u32)(a[q] >> 32), dd, 64) << 32) | __shfl_up((u32)a[q], dd, 64);
            for (int q = 0; q < 4; q++) y[q] = __shfl_up(m[q], dd, 64);
            if ((int)lane >= dd) {
                a[0] += x[0]; a[1] += x[1];
                for (int q = 0; q < 4; q++) m[q] = max_def(y[q], m[q]);
            }
        }
        if (lane == 63) { wsum[0][wave] = a[0]; wsum[1][wave] = a[1]; for (int q = 0; q < 4; q++) wdef[q][wave] = m[q]; }
        // what the lanes in front of this one (in its wave) add up to
        u32 prev[4];
        for (int q = 0; q < 4; q++) { prev[q] = __shfl_up(m[q], 1, 64); if (lane == 0) prev[q] = ZL3_NONE; }
        __syncthreads();
        u64 base[2] = {carry_sum[0], carry_sum[1]};
        u32 bdef[4] = {carry_def[0], carry_def[1], carry_def[2], carry_def[3]};
        for (u32 w = 0; w < wave; w++) {  // the waves in front of this one
            base[0] += wsum[0][w]; base[1] += wsum[1][w];
            for (int q = 0; q < 4; q++) bdef[q] = max_def(bdef[q], wdef[q][w]);
        }
        if (on) {
            u32* o = d.seg_tmp + (size_t)(s0 + t) * 16;
            for (int q = 0; q < 2; q++) {
                const u64 run = base[q] + a[q] - own[q];
                o[6 + q] = (u32)run;  // (64-bit offsets: low word, high word)
                o[12 + q] = (u32)(run >> 32);
            }
            for (int q = 0; q < 4; q++) o[8 + q] = max_def(bdef[q], prev[q]);  // words 8..11
        }
        __syncthreads();
        if (t == 1023) {  // (the last wave's base plus its own total: the tile's)
            carry_sum[0] = base[0] + a[0]; carry_sum[1] = base[1] + a[1];
            for (int q = 0; q < 4; q++) carry_def[q] = max_def(bdef[q], m[q]);
        }
        __syncthreads();
    }
    if (t == 0) {
        info->lit_total = carry_sum[0];
        info->seq_total = carry_sum[1];
    }
}

__global__ __launch_bounds__(256) void zl3_plan_c_kernel(zl3_dev d, zl_frame_desc fd) {
    zl3_info* info = d.info;
    if (info->err) return;
    const u32 ns = info->nsegs;
    // ---- C: segment records
    {
        const u32 s = blockIdx.x * 256 + threadIdx.x;
        if (s >= ns) return;
        const u32* o = d.seg_tmp + (size_t)s * 16;
        const u32 b0 = d.seg_first[s], b1 = d.seg_first[s + 1];
        zl2_frame_plan p;
        p.lit_off = (u64)o[6] | ((u64)o[12] << 32);
        p.seq_off = (u64)o[7] | ((u64)o[13] << 32);
        p.lit_cap = o[0]; p.seq_cap = o[1]; p.route = 1; p.pad = 0;
        d.plan[s] = p;
        zl2_frame_rec fr;
        fr.err = 0; fr.nblocks = b1 - b0; fr.has_checksum = 0; fr.skippable = 0;
        fr.fcs = ZL_CONTENTSIZE_UNKNOWN; fr.end_pos = 0; fr.cost = o[1]; fr.lit_cost = o[0]; fr.pad2 = 0;
        d.frec[s] = fr;
        d.frames[s] = fd;
        // inherited tables; every repeat inside the segment must have something to repeat (zstd.c:43171, :43666)
        u32 cur[4] = {o[8], o[9], o[10], o[11]};  // latest defining block in front of the segment: Huffman tree, LL, OF, ML
        zl2_blk_rec none;
        none.type = ZL3_NONE;
        zl2_blk_rec* x = d.ext + (size_t)s * 4;
        x[0] = none; x[1] = none; x[2] = none; x[3] = none;
        if (cur[0] != ZL3_NONE) { x[0] = d.braw[cur[0]]; x[0].pad[0] = 0; }  // (pad[0] of a donor record: 1 = the dictionary's, zl_decode2.hip ext_src; a block's own record keeps a byte count there)
        else if (d.dict_ext) { x[0] = d.dict_ext[0]; x[0].pad[0] = 1; }  // (the dictionary's Huffman table is in force in front of the segment)
        {   // the (up to three) distinct defining blocks of the sequence tables, in block order
            u32 a0 = cur[1], a1 = cur[2], a2 = cur[3];
            const u32 nbk = info->nblocks;
            // (frame order: a run's donors, indices nblocks .., come in front of its own blocks; ZL3_NONE sorts last)
            #define ZL3_ORDER_KEY(i_) ((i_) == ZL3_NONE ? ZL3_NONE : ((i_) >= nbk ? (i_) - nbk : (i_) + 4u))
            if (ZL3_ORDER_KEY(a0) > ZL3_ORDER_KEY(a1)) { const u32 q = a0; a0 = a1; a1 = q; }
            if (ZL3_ORDER_KEY(a1) > ZL3_ORDER_KEY(a2)) { const u32 q = a1; a1 = a2; a2 = q; }
            if (ZL3_ORDER_KEY(a0) > ZL3_ORDER_KEY(a1)) { const u32 q = a0; a0 = a1; a1 = q; }
            #undef ZL3_ORDER_KEY
            u32 k = 1;
            // (a table nobody has described yet is the dictionary's: its record spells out all three and comes first, the frame's own
            // donors -- two at most, then -- override what they define)
            if (d.dict_ext && (cur[1] == ZL3_NONE || cur[2] == ZL3_NONE || cur[3] == ZL3_NONE)) { x[k] = d.dict_ext[1]; x[k].pad[0] = 1; k++; }
            if (a0 != ZL3_NONE) { x[k] = d.braw[a0]; x[k++].pad[0] = 0; }
            if (a1 != ZL3_NONE && a1 != a0) { x[k] = d.braw[a1]; x[k++].pad[0] = 0; }
            if (a2 != ZL3_NONE && a2 != a1) { x[k] = d.braw[a2]; x[k++].pad[0] = 0; }
        }
        for (u32 b = b0; b < b1; b++) {
            const zl2_blk_rec* r = d.braw + b;
            if (r->type != 2) continue;
            // (with a formatted dictionary in the context such a block rides on the DICTIONARY's tables, zstd.c:42053: x[] above names
            // them when the context brought their records; without those the general kernel takes the frame)
            if (r->lit_mode == 3 && cur[0] == ZL3_NONE && !d.dict_ext) info_err(info, d.dict_tables ? ZL_E_FALLBACK : ZL_E_DICT_CORRUPTED);
            if (r->lit_mode == 2) cur[0] = b;
            if (r->nseq > 0 && r->seq_modes != 0x100u) {
                const u32 m = r->seq_modes;
                const u32 mk[3] = {m >> 6, (m >> 4) & 3, (m >> 2) & 3};
                for (int q = 0; q < 3; q++) {
                    if (mk[q] == 3) { if (cur[1 + q] == ZL3_NONE && !d.dict_ext) info_err(info, d.dict_tables ? ZL_E_FALLBACK : ZL_E_CORRUPTION); }
                    else cur[1 + q] = b;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// after the entropy kernels: output position and starting repcode history of every block.  One wave, 64
// blocks at a time, both as prefix scans over the lanes.  The history a block leaves behind is a function of
// the history it started with: each of the three offsets is a number, or "starting offset j minus k" (the
// symbols the sequence lanes write, zl_decode2.hip ZL2_REP_SYM).  Such functions compose into functions
// of the same shape, so the chain through all blocks (zstd.c:44291: every sequence may touch the history)
// is a scan with that composition, not a walk: round 1 walked it on one lane, 0.2 us per block -- 10 ms
// for the 51 200 blocks of a 1 GiB frame of this library's own 16 KiB blocks.
// ---------------------------------------------------------------------------------------
#define ZL3_RESOLVE(v_, r0_, r1_, r2_) (((v_) >> 31) ? (((v_) & 3u) == 0 ? (r0_) : (((v_) & 3u) == 1 ? (r1_) : (r2_))) - (((v_) & 0x7FFFFFFFu) >> 2) : (v_))
// x applied after g = (g0, g1, g2): what x means in terms of the history g started from
__device__ __forceinline__ u32 rep_compose(u32 x, u32 g0, u32 g1, u32 g2) {
    if (!(x >> 31)) return x;
    const u32 j = x & 3u, k = (x & 0x7FFFFFFFu) >> 2;
    const u32 g = j == 0 ? g0 : (j == 1 ? g1 : g2);
    if (!(g >> 31)) return g - k;
    return 0x80000000u | (g & 3u) | ((((g & 0x7FFFFFFFu) >> 2) + k) << 2);
}
// lane per block: what the block produces and the history it leaves, gathered from the sequence lanes' records
// (four dependent loads per block: done by all blocks at once here, not inside the scan)
__global__ __launch_bounds__(256) void zl3_totals_kernel(zl3_dev d, const u32* seq_scratch, uint4* blk_tot) {
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    if (d.info->err || b >= d.info->nblocks) return;
    const u32 slot = d.blk_slot[b];
    const zl2_blk_rec r = d.brec[slot];
    uint4 o = make_uint4(0, 0x80000000u, 0x80000001u, 0x80000002u);  // a block without sequences leaves the history as it was
    if (r.type != 2) o.x = r.size;
    else if (r.nseq == 0) o.x = r.lit_size;
    else {
        const uint4* q = (const uint4*)seq_scratch + d.plan[slot / ZL2_MAXB].seq_off + r.seq_dst;
        const uint4 tot = q[r.nseq], fh = q[r.nseq + 1];
        o.x = tot.y + (r.lit_size - tot.x);
        o.y = fh.x; o.z = fh.y; o.w = fh.z;
    }
    blk_tot[b] = o;
}

// (round 4: sixteen waves instead of one.  One wave took 1.17 ms for the 65 536 blocks of this library's 1 GiB frame -- a
// twelfth of its decode, and 0.2 ms of every run of a frame decoded in runs.  Every wave now scans a contiguous share of the
// blocks twice: once for its share's size and history FUNCTION (the same composition, kept symbolic), then -- the shares' sums
// and functions folded into starting points by one lane -- again for the values it writes.)
#define ZL3_OFF_WAVES 16
__global__ __launch_bounds__(64 * ZL3_OFF_WAVES) void zl3_offsets_kernel(zl3_dev d, zl_frame_desc fd, const uint4* blk_tot, u32 rep_a, u32 rep_b, u32 rep_c) {
    __shared__ u32 s_first[ZL3_OFF_WAVES];
    __shared__ u64 s_sz[ZL3_OFF_WAVES];
    __shared__ u32 s_f[ZL3_OFF_WAVES][3], s_r[ZL3_OFF_WAVES + 1][3], s_bad[ZL3_OFF_WAVES];
    __shared__ u64 s_op[ZL3_OFF_WAVES + 1];
    zl3_info* info = d.info;
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6, nb = info->nblocks, ns = info->nsegs;
    if (info->err) return;  // (uniform: nobody has written it since the launch)
    // the segments' errors first (lowest segment wins, as the reference would meet it first)
    {
        u32 first = ZL3_NONE;
        for (u32 s = t; s < ns; s += 64 * ZL3_OFF_WAVES) if (d.frec[s].err && s < first) first = s;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) { const u32 o = __shfl_xor(first, k, 64); first = o < first ? o : first; }
        if (lane == 0) s_first[wave] = first;
        __syncthreads();
        first = s_first[0];
        for (u32 w = 1; w < ZL3_OFF_WAVES; w++) first = s_first[w] < first ? s_first[w] : first;
        if (first != ZL3_NONE) { if (t == 0) info->err = d.frec[first].err; return; }
    }
    const u32 tiles = (nb + 63) / 64, per = (tiles + ZL3_OFF_WAVES - 1) / ZL3_OFF_WAVES;
    const u32 b_lo = wave * per * 64, b_hi = (wave + 1) * per * 64 < nb ? (wave + 1) * per * 64 : nb;
    // one tile of 64 blocks: inclusive scans over the lanes -- sizes by addition, histories by composition (the earlier block's function first)
    auto scan_tile = [&](u32 b0, u32& sz, u32& acc, u32& f0, u32& f1, u32& f2, bool& over) {
        const u32 b = b0 + lane;
        sz = 0;
        f0 = 0x80000000u; f1 = 0x80000001u; f2 = 0x80000002u;
        if (b < b_hi) { const uint4 o = blk_tot[b]; sz = o.x; f0 = o.y; f1 = o.z; f2 = o.w; }
        over = __ballot(sz > ZL_BLOCK_MAX) != 0;
        acc = sz;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const u32 a = __shfl_up(acc, dd, 64);
            const u32 g0 = __shfl_up(f0, dd, 64), g1 = __shfl_up(f1, dd, 64), g2 = __shfl_up(f2, dd, 64);
            if ((int)lane >= dd) {
                acc += a;
                const u32 n0 = rep_compose(f0, g0, g1, g2), n1 = rep_compose(f1, g0, g1, g2), n2 = rep_compose(f2, g0, g1, g2);
                f0 = n0; f1 = n1; f2 = n2;
            }
        }
    };
    // ---- pass 1: this wave's share as ONE function of the history in front of it, and its size
    {
        u64 op = 0;
        u32 F0 = 0x80000000u, F1 = 0x80000001u, F2 = 0x80000002u, bad = 0;
        for (u32 b0 = b_lo; b0 < b_hi; b0 += 64) {
            u32 sz, acc, f0, f1, f2;
            bool over;
            scan_tile(b0, sz, acc, f0, f1, f2, over);
            if (over) bad = ZL_E_CORRUPTION;
            const u32 t0 = __shfl(f0, 63, 64), t1 = __shfl(f1, 63, 64), t2 = __shfl(f2, 63, 64);
            const u32 n0 = rep_compose(t0, F0, F1, F2), n1 = rep_compose(t1, F0, F1, F2), n2 = rep_compose(t2, F0, F1, F2);
            F0 = n0; F1 = n1; F2 = n2;
            op += __shfl(acc, 63, 64);
        }
        if (lane == 0) { s_sz[wave] = op; s_f[wave][0] = F0; s_f[wave][1] = F1; s_f[wave][2] = F2; s_bad[wave] = bad; }
    }
    __syncthreads();
    if (t == 0) {  // where every share starts: output position and history (1, 4, 8 at the start of a frame, zstd.c:41433; a dictionary brings its own, zstd.c:42053)
        u64 op = 0;
        u32 r0 = rep_a, r1 = rep_b, r2 = rep_c;
        for (u32 w = 0; w < ZL3_OFF_WAVES; w++) {
            s_op[w] = op; s_r[w][0] = r0; s_r[w][1] = r1; s_r[w][2] = r2;
            const u32 n0 = ZL3_RESOLVE(s_f[w][0], r0, r1, r2), n1 = ZL3_RESOLVE(s_f[w][1], r0, r1, r2), n2 = ZL3_RESOLVE(s_f[w][2], r0, r1, r2);
            r0 = n0; r1 = n1; r2 = n2;
            op += s_sz[w];
        }
        s_op[ZL3_OFF_WAVES] = op; s_r[ZL3_OFF_WAVES][0] = r0; s_r[ZL3_OFF_WAVES][1] = r1; s_r[ZL3_OFF_WAVES][2] = r2;
    }
    __syncthreads();
    // ---- pass 2: the values
    {
        u64 op = s_op[wave];
        u32 r0 = s_r[wave][0], r1 = s_r[wave][1], r2 = s_r[wave][2];
        for (u32 b0 = b_lo; b0 < b_hi; b0 += 64) {
            const u32 b = b0 + lane;
            u32 sz, acc, f0, f1, f2;
            bool over;
            scan_tile(b0, sz, acc, f0, f1, f2, over);
            // what the block starts with: the blocks in front of it in this tile, applied to the history in front of the tile
            u32 e0 = __shfl_up(f0, 1, 64), e1 = __shfl_up(f1, 1, 64), e2 = __shfl_up(f2, 1, 64);
            if (lane == 0) { e0 = 0x80000000u; e1 = 0x80000001u; e2 = 0x80000002u; }
            if (b < b_hi) {
                d.blk_dst[b] = (u32)(op + acc - sz);
                d.rep_start[(size_t)b * 3 + 0] = ZL3_RESOLVE(e0, r0, r1, r2);
                d.rep_start[(size_t)b * 3 + 1] = ZL3_RESOLVE(e1, r0, r1, r2);
                d.rep_start[(size_t)b * 3 + 2] = ZL3_RESOLVE(e2, r0, r1, r2);
            }
            const u32 t0 = __shfl(f0, 63, 64), t1 = __shfl(f1, 63, 64), t2 = __shfl(f2, 63, 64);
            const u32 n0 = ZL3_RESOLVE(t0, r0, r1, r2), n1 = ZL3_RESOLVE(t1, r0, r1, r2), n2 = ZL3_RESOLVE(t2, r0, r1, r2);
            r0 = n0; r1 = n1; r2 = n2;
            op += __shfl(acc, 63, 64);
        }
    }
    if (t == 0) {
        const u64 op = s_op[ZL3_OFF_WAVES];
        u32 bad = 0;
        for (u32 w = 0; w < ZL3_OFF_WAVES; w++) bad |= s_bad[w];
        d.blk_dst[nb] = (u32)op;
        info->produced = op;
        info->rep_end[0] = s_r[ZL3_OFF_WAVES][0]; info->rep_end[1] = s_r[ZL3_OFF_WAVES][1]; info->rep_end[2] = s_r[ZL3_OFF_WAVES][2];
        if (bad) info->err = bad;
        else if (op > fd.dst_cap) info->err = ZL_E_DSTSIZE_TOO_SMALL;
        else if (info->fcs != ZL_CONTENTSIZE_UNKNOWN && op != info->fcs) info->err = ZL_E_CORRUPTION;  // zstd.c:41646
    }
}

// ---------------------------------------------------------------------------------------
// the deferred bytes.  After the exec kernel every output byte is either final (link ZL3_ROOT) or waits
// for the frame position in its link, which lies in front of its exec workgroup's run and may be waiting
// itself.  The output is worked through in GROUPS of ZL3_GROUP_BYTES, one launch per group, in order: a
// waiting byte follows links towards the front until it stands on a byte that is final -- because the
// exec kernel left it final, or because it lies in front of the group, where every byte has been made
// final by the launches before this one -- and copies it.  Nothing is written but the output bytes
// themselves (links stay as they are, so no workgroup depends on what another one of the same launch
// does), every waiting byte is visited exactly once, and a walk is short: every link leaves its exec
// run, so it takes at most as many steps as there are runs in a group.  No host round trip between the
// launches.  (Round 1 made one launch per run: 1023 launches, 8.3 ms for their number alone on the
// 1 GiB frame; pointer-jumping rounds over the whole frame instead cost more than that in link reads,
// because text-like frames chain through every run: depth 256 and more.)
// ---------------------------------------------------------------------------------------
#define ZL3_MARK_MIN (ZL3_ROOT - 2)
#ifndef ZL3_GROUP_LOG
#define ZL3_GROUP_LOG 22                       // 4 MiB of output per launch (measured 8 / 4 / 2 / 1 MiB on the 1 GiB reference frame: 7.0 / 5.8 / 6.6 / 8.5 ms)
#endif
#define ZL3_COPY_THREADS 256
#define ZL3_NO_RESET (~0ull)
#ifndef ZL3_KEEP_LINKS
#define ZL3_KEEP_LINKS 1   // 0: the link array is filled in front of every decode, as rounds 1-5
#endif
#ifndef ZL3_COPY_SLICE
#define ZL3_COPY_SLICE 1024u                   // output bytes per workgroup: four per thread (4096: every thread made four rounds one behind the other; 23.0 -> 21.8 ms on the 1 GiB reference frame)
#endif
// Round 6, measured on the 1 GiB reference frame (ordered copies 3.9 of its 19.6 ms; 192 launches of 17 - 20 us) and taken out again:
//  * with the walks and the byte reads compiled out the launches take 0.6 ms together, with the walks alone out 2.4: the time is the
//    dependent trips to memory, not the scan of the links;
//  * several quads per thread travelling TOGETHER (all their links asked for at once, walks in step; 2048 / 1024 workgroups per launch
//    instead of 4096): 1 / 2 / 4 quads 19.7 / 19.9 / 20.6 ms, this library's own frame 10.4 / 10.6 / 11.3 -- a wave walks as long as
//    the longest of its bytes' chains;
//  * a quad that waits for four consecutive positions (the inside of a match) moved by ONE 16-byte link read and ONE 4-byte read: no change;
//  * the walks of ALL groups in one launch up front (they only read and shorten links: no order among them), the ordered launches
//    left with link -> byte -> store: the ordered launches 19.7 -> 11.9 us each, but the walk launch takes 2.2 ms -- the walks are bound
//    by their number (one 64-byte line per link read of a short match), not by the order: 3.96 -> 4.5 ms.
__global__ __launch_bounds__(ZL3_COPY_THREADS) void zl3_group_copy_kernel(u8* dst, u32* link, const u32* tile_flag, u64 gstart, u64 reset_start, zl3_info* info) {
    const u64 n = info->produced;
    if (info->err) return;
    // The group that was copied by the launch in front of this one gets its links back to "final": every one of its bytes is, nothing in
    // this launch reads a link in front of gstart, and a link array that is all ZL3_ROOT when a decode ends needs no fill when the next one
    // begins (zl3_state_s::link_clean; the fill of 4 GiB for a 1 GiB frame ran beside the entropy stage and cost this library's own frame
    // 0.7 of its 10.2 ms there).
    if (reset_start != ZL3_NO_RESET) {
        const u64 r0 = reset_start + (u64)blockIdx.x * ZL3_COPY_SLICE;
        if (r0 < n && tile_flag[r0 >> ZL3_TILE_LOG]) {
            const u64 q = r0 + threadIdx.x * 4;
            if (q < n) {
                const uint4 v = *(const uint4*)(link + q);
                if (v.x < ZL3_MARK_MIN || v.y < ZL3_MARK_MIN || v.z < ZL3_MARK_MIN || v.w < ZL3_MARK_MIN) *(uint4*)(link + q) = make_uint4(ZL3_ROOT, ZL3_ROOT, ZL3_ROOT, ZL3_ROOT);
            }
        }
    }
    const u64 s0 = gstart + (u64)blockIdx.x * ZL3_COPY_SLICE;
    if (s0 >= n) return;
    if (!tile_flag[s0 >> ZL3_TILE_LOG]) return;  // (a slice lies inside one 16 KiB tile)
    for (u64 q = s0 + threadIdx.x * 4; q < s0 + ZL3_COPY_SLICE && q < n; q += ZL3_COPY_THREADS * 4) {
        const uint4 v = *(const uint4*)(link + q);  // (the link buffer is padded: reading a few entries past the end is fine)
        const u32 lk[4] = {v.x, v.y, v.z, v.w};
        if (lk[0] >= ZL3_MARK_MIN && lk[1] >= ZL3_MARK_MIN && lk[2] >= ZL3_MARK_MIN && lk[3] >= ZL3_MARK_MIN) continue;
        u32 word = 0;
        if (q + 4 <= n) __builtin_memcpy(&word, dst + q, 4);  // (the frame may start at any address)
        else for (int j = 0; j < 4 && q + j < n; j++) word |= (u32)dst[q + j] << (8 * j);
        // the four bytes' walks step together: their link reads travel side by side (one byte after the other, a thread's
        // chain of dependent trips to memory was four times as long -- and the launches of a frame run one behind the other)
        u32 sj[4], act = 0, walking = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            sj[j] = lk[j];
            if (q + j >= n || sj[j] >= ZL3_MARK_MIN) continue;
            if ((u64)sj[j] >= q + j) { info_err(info, ZL_E_CORRUPTION); continue; }  // (links only ever point backwards; a guard against garbage)
            act |= 1u << j;
            if ((u64)sj[j] >= gstart) walking |= 1u << j;
        }
        while (walking) {
            u32 l[4];
#pragma unroll
            for (int j = 0; j < 4; j++) l[j] = ((walking >> j) & 1) ? link[sj[j]] : 0u;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!((walking >> j) & 1)) continue;
                if (l[j] >= ZL3_MARK_MIN) walking &= ~(1u << j);                                  // final since the exec kernel
                else if (l[j] >= sj[j]) { info_err(info, ZL_E_CORRUPTION); walking &= ~(1u << j); }
                else { sj[j] = l[j]; if ((u64)sj[j] < gstart) walking &= ~(1u << j); }
            }
        }
        u32 bj[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bj[j] = ((act >> j) & 1) ? (u32)dst[sj[j]] : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) if ((act >> j) & 1) word = (word & ~(0xFFu << (8 * j))) | (bj[j] << (8 * j));
        if (q + 4 <= n) __builtin_memcpy(dst + q, &word, 4);
        else for (int j = 0; j < 4 && q + j < n; j++) dst[q + j] = (u8)(word >> (8 * j));
    }
}

// ---------------------------------------------------------------------------------------
// the frame's result: checksum and size (zstd.c:41646-41660)
// ---------------------------------------------------------------------------------------
// (have_hash: the host has run the XXH64 chain over the regenerated bytes -- zl3_hash_on_host says why -- and hands its low 32 bits in)
// which groups of output hold a byte that waits at all (zl3_decode_frame launches the ordered copies of those only)
__global__ __launch_bounds__(256) void zl3_group_flags_kernel(const u32* tile_flag, u32* group_flag, u32 tiles_per_group, u64 ntiles) {
    __shared__ u32 any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const u64 t0 = (u64)blockIdx.x * tiles_per_group;
    u32 mine = 0;
    for (u32 i = threadIdx.x; i < tiles_per_group; i += 256) if (t0 + i < ntiles && tile_flag[t0 + i]) mine = 1;
    if (mine) any = 1;
    __syncthreads();
    if (threadIdx.x == 0) group_flag[blockIdx.x] = any;
}

__global__ __launch_bounds__(512) void zl3_finish_kernel(zl3_info* info, zl_frame_desc fd, const u8* src_base, const u8* dst_base, u64* result, u32 ignore_checksum, u32 have_hash, u32 hash) {
    __shared__ __align__(16) u8 tile[8192];
    __shared__ u64 acc[4];
    __shared__ u32 bc[4];
    u32 err = info->err;
    if (!err && info->pending) err = ZL_E_CORRUPTION;  // (cannot happen for links that only point backwards; kept as a guard)
    if (!err && info->has_checksum && !ignore_checksum) {
        const u32 got = have_hash ? hash : zl_xxh64_low32(tile, acc, bc, dst_base + fd.dst_off, info->produced);
        if (got != zl_rd_le(src_base + fd.src_off + info->end_pos, 4)) err = ZL_E_CHECKSUM_WRONG;
    }
    if (threadIdx.x == 0) *result = err ? ZL_ERR(err) : info->produced;
}

// the result word of a frame decoded in runs: its content checksum over the whole (zl3_check_content)
__global__ __launch_bounds__(512) void zl3_check_kernel(const u8* content, u64 n, const u8* expect, u64* result, u32 have_hash, u32 hash) {
    __shared__ __align__(16) u8 tile[8192];
    __shared__ u64 acc[4];
    __shared__ u32 bc[4];
    u32 err = 0;
    if (expect) {
        const u32 got = have_hash ? hash : zl_xxh64_low32(tile, acc, bc, content, n);
        if (got != zl_rd_le(expect, 4)) err = ZL_E_CHECKSUM_WRONG;
    }
    if (threadIdx.x == 0) *result = err ? ZL_ERR(err) : n;
}

}  // namespace


#define ZL3_SINK_LOG 26      // host entries: the output is downloaded in pieces of 64 MiB behind the copy stage (zl3_hint::sink_host)
#define ZL3_SINK_EVENTS 64
struct zl3_state_s {
    DevBuf info, hblk, braw, seg_first, blk_slot, brec, frec, plan, frames, ext, blk_dst, rep_start, seg_tmp, order, word, lit, seq, link, tile_flag, big, cnt, first, tot, spec, gflag;
    PinBuf h_spec, h_hash, h_gflag;
    hipEvent_t sink_ev[ZL3_SINK_EVENTS] = {};
    hipEvent_t sink_done[ZL3_SINK_EVENTS] = {};  // behind the download of each piece (zl3_hint::sink_hash)
    hipEvent_t fill_fork = nullptr, fill_join = nullptr;  // around the link fill that runs beside the entropy stage
    bool link_clean = false;   // the first link_clean_bytes of the link array hold ZL3_ROOT throughout: the decode before this one ended well and its ordered copies put every link back
    size_t link_clean_bytes = 0;
    // what phase 1 of a run in two calls leaves for phase 2 (zl3_hint::phase)
    struct Saved { zl3_dev d; zl2_args a; zl3_info info; u64 span = 0; hipStream_t fill_st = nullptr; bool valid = false; } sv;
};

static bool zl3_link_ensure(zl3_state_s* s, size_t bytes) {
    const void* const p0 = s->link.p;
    const size_t c0 = s->link.cap;
    if (!s->link.ensure(bytes)) { s->link_clean = false; s->link_clean_bytes = 0; return false; }
    if (s->link.p != p0 || s->link.cap != c0) { s->link_clean = false; s->link_clean_bytes = 0; }  // (new memory: nothing is known about it)
    return true;
}
extern "C" zl3_state* zl3_create(void) { return new (std::nothrow) zl3_state_s(); }
extern "C" void zl3_free(zl3_state* s) {
    if (!s) return;
    for (DevBuf* b : {&s->info, &s->hblk, &s->braw, &s->seg_first, &s->blk_slot, &s->brec, &s->frec, &s->plan, &s->frames, &s->ext, &s->blk_dst, &s->rep_start, &s->seg_tmp, &s->order, &s->word,
                      &s->lit, &s->seq, &s->link, &s->tile_flag, &s->big, &s->cnt, &s->first, &s->tot, &s->spec, &s->gflag}) b->release();
    s->h_spec.release();
    s->h_gflag.release();
    s->h_hash.release();
    for (hipEvent_t e : s->sink_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->sink_done) if (e) (void)hipEventDestroy(e);
    if (s->fill_fork) (void)hipEventDestroy(s->fill_fork);
    if (s->fill_join) (void)hipEventDestroy(s->fill_join);
    delete s;
}

// The content checksum of a large frame whose bytes lie in DEVICE memory (the device-resident entries; the host entries check
// theirs where the download put them).  XXH64 is one chain through the whole content (zl_xxh64.h): four GPU lanes run it at
// 0.5 GB/s -- 2 s for a frame of 1 GiB that was decoded in 14 ms, which is what this pipeline did until the end of round 3
// (scripts/dev/checksum_cost.py) -- a CPU core at 15 GB/s and more.  So the bytes come down in pieces through two pinned buffers,
// the copy of one piece behind the hashing of the one before, as zl_compress_device_finish does for the same reason.
static bool zl3_hash_on_host(zl3_state* s, const u8* d_ptr, u64 n, hipStream_t st, u32* low32) {
    const size_t piece = (size_t)16 << 20;
    if (!s->h_hash.ensure(2 * piece)) return false;
    zlx::Xxh64 h;
    u64 sent = 0;
    size_t k = 0;
    auto fetch = [&](size_t slot) -> size_t {
        const size_t len = (size_t)std::min<u64>(piece, n - sent);
        if (hipMemcpyAsync((u8*)s->h_hash.p + slot * piece, d_ptr + sent, len, hipMemcpyDeviceToHost, st) != hipSuccess) return 0;
        sent += len;
        return len;
    };
    size_t len_cur = n ? fetch(0) : 0;
    if (n && !len_cur) return false;
    while (len_cur) {
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        size_t len_next = 0;
        if (sent < n) { len_next = fetch((k + 1) & 1); if (!len_next) return false; }
        h.update((const u8*)s->h_hash.p + (k & 1) * piece, len_cur);
        len_cur = len_next;
        k++;
    }
    if (h.total != n) return false;
    *low32 = h.low32();
    return true;
}

extern "C" hipError_t zl3_check_content(zl3_state* s, const u8* d_content, u64 n, const u8* d_expect, u64* d_result, hipStream_t st) {
    u32 hash = 0;
    if (d_expect && (!s || !zl3_hash_on_host(s, d_content, n, st, &hash))) return hipErrorUnknown;
    hipLaunchKernelGGL(zl3_check_kernel, dim3(1), dim3(512), 0, st, d_content, n, d_expect, d_result, d_expect ? 1u : 0u, hash);
    return hipGetLastError();
}

// Decodes one large frame.  Returns 0 (the result word of the frame has been written on the device),
// ZL_ERR(ZL_E_FALLBACK) when the frame has to take the general kernel, or another error of the launch itself.
extern "C" size_t zl3_decode_frame(zl3_state* s, const zl_frame_desc* fdp, const u8* d_src, u8* d_dst, u64* d_result, int ignore_checksum, hipStream_t st, const zl2_side* side, const zl3_hint* hint) {
    zl_frame_desc fd = *fdp;
    const u32 phase = hint ? hint->phase : 0u;
    if (phase == 1) fd.dst_off = 0;  // (the segments' descriptors carry it: phase 2 moves the output pointer instead)
#ifdef ZL_DEV_SWITCHES  // development builds: ZL_DEBUG_BIG=1 prints the time of every stage to stderr (each stamp synchronises the stream)
    static const bool dbg = getenv("ZL_DEBUG_BIG") != nullptr;
    auto now = [&]() { (void)hipStreamSynchronize(st); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = dbg ? now() : 0;
    auto stamp = [&](const char* what) { if (dbg) { const double t1 = now(); fprintf(stderr, "[zl3] %-10s %8.3f ms\n", what, t1 - t0); t0 = t1; } };
#else
    auto stamp = [](const char*) {};
#endif
    zl3_dev d;
    zl3_info info;
    zl2_args a;
    u64 span = 0;
    hipStream_t fill_st = st;
    const bool have_dict = hint && hint->dict_len != 0;
    if (phase == 2) {  // the second call of a run: everything up to the entropy stage has been queued by the first
        if (!s->sv.valid) return ZL_ERR(ZL_E_GENERIC);
        s->sv.valid = false;
        d = s->sv.d; a = s->sv.a; info = s->sv.info; span = s->sv.span; fill_st = s->sv.fill_st;
        fd.dst_off = 0;
        d_dst += fdp->dst_off;
    }
    if (phase != 2) {
    if (fd.src_size >= 0xFFFFFF00ull) return ZL_ERR(ZL_E_FALLBACK);  // block records are 32-bit
    if (!s->fill_fork && (hipEventCreateWithFlags(&s->fill_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->fill_join, hipEventDisableTiming) != hipSuccess)) { s->fill_fork = s->fill_join = nullptr; }
    // (dst_cap may be the room left in the caller's buffer rather than the frame's size: what the frame can produce is bounded by its input too)
    const u64 out_bound = std::min<u64>(fd.dst_cap, fd.src_size * 1024 + ((u64)1 << 20));
    // blocks planned for: full blocks plus a generous allowance for encoders that split blocks (this library's own cut 16 KiB blocks)
    u32 cap_blocks = (u32)std::min<u64>((out_bound >> 13) + (fd.src_size >> 11) + 64, 1u << 22) & ~(u32)(ZL2_MAXB - 1);
    if (hint && hint->host_blocks && hint->nblocks + 4 > cap_blocks) cap_blocks = (hint->nblocks + 4 + ZL2_MAXB) & ~(u32)(ZL2_MAXB - 1);
    if (hint && hint->reserve_blocks + 4 > cap_blocks) cap_blocks = (hint->reserve_blocks + 4 + ZL2_MAXB) & ~(u32)(ZL2_MAXB - 1);
    const u32 cap_segs = cap_blocks;  // (a segment holds at least one block)
    if (!s->info.ensure(sizeof(zl3_info)) || !s->hblk.ensure((size_t)cap_blocks * sizeof(zl3_hblk)) || !s->braw.ensure((size_t)cap_blocks * sizeof(zl2_blk_rec)) ||
        !s->seg_first.ensure(((size_t)cap_segs + 1) * 4) || !s->blk_slot.ensure((size_t)cap_blocks * 4) || !s->blk_dst.ensure(((size_t)cap_blocks + 1) * 4) ||
        !s->rep_start.ensure((size_t)cap_blocks * 12) || !s->word.ensure(64) || !s->big.ensure(sizeof(zl3_big)))
        return ZL_OOM_HERE;
    d.info = (zl3_info*)s->info.p; d.hblk = (zl3_hblk*)s->hblk.p; d.braw = (zl2_blk_rec*)s->braw.p; d.seg_first = (u32*)s->seg_first.p; d.blk_slot = (u32*)s->blk_slot.p;
    d.blk_dst = (u32*)s->blk_dst.p; d.rep_start = (u32*)s->rep_start.p; d.max_blocks = (u32*)s->word.p; d.cap_blocks = cap_blocks;
    d.have_dict = have_dict ? 1u : 0u;
    d.dict_id = have_dict ? hint->dict_id : 0u;
    d.dict_tables = (have_dict && hint->dict_formatted) ? 1u : 0u;
    d.dict_ext = d.dict_tables ? hint->dict_ext : nullptr;
    d.brec = nullptr; d.frec = nullptr; d.plan = nullptr; d.frames = nullptr; d.ext = nullptr; d.seg_tmp = nullptr;
    if (hipMemsetAsync(d.info, 0, sizeof(zl3_info), st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    // ---- the block table: uploaded from the host's walk, walked per chunk with the caller's chunk offsets, or walked by one lane
    bool walked = false;
    if (hint && hint->host_blocks && hint->nblocks) {
        if (hipMemcpyAsync(d.hblk, hint->host_blocks, (size_t)hint->nblocks * sizeof(zl3_hblk), hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        const u32 nd = hint->headerless ? std::min<u32>(hint->ndonors, 4u) : 0u;
        if (nd && hipMemcpyAsync(d.hblk + hint->nblocks, hint->host_blocks + hint->nblocks, (size_t)nd * sizeof(zl3_hblk), hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        hipLaunchKernelGGL(zl3_host_walk_kernel, dim3(1), dim3(64), 0, st, d, fd, d_src, hint->nblocks, hint->end_pos, hint->headerless, hint->size_unknown, nd, make_uint4(hint->donor[0], hint->donor[1], hint->donor[2], hint->donor[3]));
        walked = true;
    } else if (hint && hint->d_chunk_off && hint->nchunks) {
        const u32 nc = hint->nchunks;
        if (!s->cnt.ensure((size_t)nc * 4) || !s->first.ensure((size_t)nc * 4)) return ZL_OOM_HERE;
        if (hipMemsetAsync(s->cnt.p, 0, (size_t)nc * 4, st) != hipSuccess || hipMemsetAsync(s->first.p, 0, (size_t)nc * 4, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        hipLaunchKernelGGL(zl3_chunk_walk_kernel, dim3((nc + 255) / 256), dim3(256), 0, st, d, fd, d_src, hint->d_chunk_off, nc, (u32*)s->cnt.p, (const u32*)s->first.p, 0u, hint->headerless, hint->not_last, hint->tail);
        hipLaunchKernelGGL(zl3_chunk_scan_kernel, dim3(1), dim3(1024), 0, st, d, fd, (const u32*)s->cnt.p, (u32*)s->first.p, nc, hint->d_chunk_off, d_src, hint->headerless, hint->tail);
        hipLaunchKernelGGL(zl3_chunk_walk_kernel, dim3((nc + 255) / 256), dim3(256), 0, st, d, fd, d_src, hint->d_chunk_off, nc, (u32*)s->cnt.p, (const u32*)s->first.p, 1u, hint->headerless, hint->not_last, hint->tail);
        zl3_info probe;
        if (hipMemcpyAsync(&probe, d.info, sizeof(probe), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (probe.err == ZL_E_FALLBACK_WALK) { if (hipMemsetAsync(d.info, 0, sizeof(zl3_info), st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC); }  // the hint does not describe this frame: walk it
        else walked = true;
    }
    if (!walked && hint && hint->headerless) return ZL_ERR(ZL_E_FALLBACK);  // (a run of blocks without a header has no other way in)
    if (!walked && fd.src_size >= 2ull * ZL3_SPEC_PART) {
        // the speculative parallel walk (above); whatever it cannot link up is walked by one lane below
        const u32 nparts = (u32)((fd.src_size + ZL3_SPEC_PART - 1) / ZL3_SPEC_PART);
        const size_t list_bytes = (size_t)nparts * ZL3_SPEC_LIST * sizeof(zl3_spec_rec), cnt_bytes = (size_t)nparts * 4;
        const size_t total = cnt_bytes + list_bytes + 2 * cnt_bytes;  // counts | lists | start | first
        if (!s->spec.ensure(total + 64) || !s->h_spec.ensure(total + sizeof(zl3_info) + 64)) return ZL_OOM_HERE;
        u32* d_counts = (u32*)s->spec.p;
        zl3_spec_rec* d_lists = (zl3_spec_rec*)((u8*)s->spec.p + cnt_bytes);
        u32* d_start = (u32*)((u8*)s->spec.p + cnt_bytes + list_bytes);
        u32* d_first = d_start + nparts;
        u8* h = (u8*)s->h_spec.p;
        if (hipMemsetAsync(d_counts, 0, cnt_bytes, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        hipLaunchKernelGGL(zl3_spec_walk_kernel, dim3((ZL3_SPEC_CANDS + 255) / 256, nparts), dim3(256), 0, st, d, fd, d_src, nparts, d_counts, d_lists);
        if (hipMemcpyAsync(h, s->spec.p, cnt_bytes + list_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(h + total, d.info, sizeof(zl3_info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        const u32* counts = (const u32*)h;
        const zl3_spec_rec* lists = (const zl3_spec_rec*)(h + cnt_bytes);
        u32* start = (u32*)(h + cnt_bytes + list_bytes);
        u32* first = start + nparts;
        zl3_info hi;
        memcpy(&hi, h + total, sizeof(hi));
        bool ok = hi.err == 0 && hi.hdr_size != 0;
        u64 cur = hi.hdr_size;
        u32 nb = 0, end_pos = 0;
        bool ended = false;
        for (u32 k = 0; k < nparts && ok; k++) {
            start[k] = 0xFFFFFFFFu; first[k] = 0;
            if (ended || cur >= ((u64)k + 1) * ZL3_SPEC_PART) continue;  // no block starts inside this part
            const u32 cnt = counts[k];
            if (cnt > ZL3_SPEC_LIST) { ok = false; break; }
            const zl3_spec_rec* hit = nullptr;
            for (u32 i = 0; i < cnt; i++) if (lists[(size_t)k * ZL3_SPEC_LIST + i].start == (u32)cur) { hit = &lists[(size_t)k * ZL3_SPEC_LIST + i]; break; }
            if (!hit) { ok = false; break; }
            start[k] = hit->start; first[k] = nb;
            nb += hit->count;
            cur = hit->exit;
            if (hit->last) { ended = true; end_pos = hit->exit; }
        }
        ok = ok && ended;
        if (ok) {
            if (hipMemcpyAsync(d_start, start, 2 * cnt_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            hipLaunchKernelGGL(zl3_spec_fill_kernel, dim3((nparts + 63) / 64), dim3(64), 0, st, d, fd, d_src, nparts, (const u32*)d_start, (const u32*)d_first, nb, end_pos);
            walked = true;
        } else if (hipMemsetAsync(d.info, 0, sizeof(zl3_info), st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);  // start over: the one-lane walk
    }
    if (!walked) hipLaunchKernelGGL(zl3_walk_kernel, dim3(1), dim3(64), 0, st, d, fd, d_src);
    hipLaunchKernelGGL(zl3_parse_kernel, dim3((cap_blocks + 255) / 256), dim3(256), 0, st, d, fd, d_src);
    hipLaunchKernelGGL(zl3_segment_kernel, dim3(1), dim3(1024), 0, st, d);
    if (hipMemcpyAsync(&info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    stamp("walk+parse");
    if (info.err == ZL_E_FALLBACK) return ZL_ERR(ZL_E_FALLBACK);
    if (!info.err) {
        const size_t ns = (hint && hint->reserve_blocks) ? std::max<size_t>(info.nsegs, (hint->reserve_blocks + ZL2_MAXB - 1) / ZL2_MAXB + 64) : (size_t)info.nsegs;  // (a segment of a run holds ZL2_MAXB blocks, or one that is large)
        if (!s->brec.ensure(ns * ZL2_MAXB * sizeof(zl2_blk_rec)) || !s->frec.ensure(ns * sizeof(zl2_frame_rec)) || !s->plan.ensure(ns * sizeof(zl2_frame_plan)) ||
            !s->frames.ensure(ns * sizeof(zl_frame_desc)) || !s->ext.ensure(ns * 4 * sizeof(zl2_blk_rec)) || !s->seg_tmp.ensure(ns * 64) || !s->order.ensure(ns * 8 + 64))
            return ZL_OOM_HERE;
        d.brec = (zl2_blk_rec*)s->brec.p; d.frec = (zl2_frame_rec*)s->frec.p; d.plan = (zl2_frame_plan*)s->plan.p; d.frames = (zl_frame_desc*)s->frames.p;
        d.ext = (zl2_blk_rec*)s->ext.p; d.seg_tmp = (u32*)s->seg_tmp.p;
        const u32 plan_wgs = info.nsegs ? (info.nsegs + 255) / 256 : 1u;
        hipLaunchKernelGGL(zl3_plan_a_kernel, dim3(plan_wgs), dim3(256), 0, st, d);
        hipLaunchKernelGGL(zl3_plan_b_kernel, dim3(1), dim3(1024), 0, st, d);
        hipLaunchKernelGGL(zl3_plan_c_kernel, dim3(plan_wgs), dim3(256), 0, st, d, fd);
        if (hipMemcpyAsync(&info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stamp("plan");
        if (info.err == ZL_E_FALLBACK) return ZL_ERR(ZL_E_FALLBACK);  // (blocks that ride on a dictionary's entropy tables)
    }
#ifdef ZL_DEV_SWITCHES
    if (dbg) fprintf(stderr, "[zl3] blocks %u segments %u literals %llu sequences %llu err %u\n", info.nblocks, info.nsegs, (unsigned long long)info.lit_total, (unsigned long long)info.seq_total, info.err);
#endif
    // output positions the links must cover: the recorded size, else what the blocks can produce at most (and the caller has room for)
    span = info.fcs != ZL_CONTENTSIZE_UNKNOWN ? info.fcs : std::min<u64>(fd.dst_cap, (u64)info.nblocks * ZL_BLOCK_MAX);
    if (span >= 0xFFFFD000ull) return ZL_ERR(ZL_E_FALLBACK);  // links are 32-bit (and the exec kernel's in-batch form of one adds its batch size to a position)
    }  // (phase != 2)
    const u64 ntiles = (span >> ZL3_TILE_LOG) + 1;
    if (!info.err) {
        if (phase != 2) {
        const u64 rspan = std::max<u64>(span, hint ? std::min<u64>(hint->reserve_span, 0xFFFFD000ull) : 0);
        // (the sequence records of later runs are not known yet: room for half as many again)
        if (!s->lit.ensure(std::max<u64>(info.lit_total, hint && hint->reserve_span ? rspan : 0) + 64) || !s->seq.ensure((hint && hint->reserve_span && s->seq.cap < info.seq_total * 16 + 64 ? info.seq_total * 24 : info.seq_total * 16) + 64) ||
            !zl3_link_ensure(s, rspan * 4 + 64) || !s->tile_flag.ensure(((rspan >> ZL3_TILE_LOG) + 1) * 4) || !s->tot.ensure((size_t)d.cap_blocks * 16))
            return ZL_OOM_HERE;
        }
        // An exec workgroup works through a RUN of consecutive segments; what a run's bytes copy from in front of the run waits for
        // the ordered copies.  Eight segments per run keep those few on a frame of thousands of blocks -- and leave a frame of a few
        // hundred with a few dozen workgroups, each as long as its eight blocks (0.75 ms a block): a 64 MiB reference frame spent
        // 6 of its 16 ms there.  The run shrinks with the frame so that the launch still fills the device (three workgroups per CU).
        // (segments of several small blocks -- this library's own frames, whose matches rarely leave their chunk -- want shorter runs
        // still: 1 GiB of them, runs of 1 / 2 / 3 / 4 / 5 / 8 segments: 12.5 / 11.3 / 10.4 / 10.8 / 11.5 / 12.1 ms; the reference's
        // single-block segments: 27.7 / 23.5 / 22.3 / 22.4 / 23.5 / 20.9 ms, best at 6 - 8)
        const u32 run_cap = info.nblocks >= 4u * info.nsegs ? 3u : (u32)ZL3_EXEC_RUN;
        // (round 6: segments of several small blocks fill the device at TWO workgroups per CU -- the runs of 1024 chunks that a frame from host memory is decoded in take two
        // chunks per workgroup instead of one: half the bytes that cross a run's edge and wait for the ordered copies.  1 GiB host to host, runs of 1 / 2 / 3 / 4: 29.0 / 27.5 / 28.5 / 29.1 ms)
        u32 exec_run = std::max<u32>(1u, std::min<u32>(run_cap, info.nsegs / ((run_cap == 3u ? 2u : 3u) * (u32)g_num_cu)));
#ifdef ZL_DEV_SWITCHES
        if (getenv("ZL_EXEC_RUN")) exec_run = (u32)atoi(getenv("ZL_EXEC_RUN"));
#endif
        if (phase != 1) {  // (the history is phase 2's to say)
            zl3_big big;
            big.info = d.info; big.link = (u32*)s->link.p; big.tile_flag = (u32*)s->tile_flag.p; big.blk_dst = d.blk_dst; big.rep_start = d.rep_start; big.seg_first = d.seg_first; big.exec_run = exec_run;
            big.dict_len = have_dict ? hint->dict_len : 0u; big.dict_end = have_dict ? hint->dict_end : nullptr;
            if (hipMemcpyAsync(s->big.p, &big, sizeof(big), hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        }
        if (phase != 2) {
        if (hipMemsetAsync(s->tile_flag.p, 0, ntiles * 4, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        // every byte starts out final; the exec kernel writes the links of the waiting ones only (a plain fill at memory speed
        // instead of one more 4-byte store per output byte from the exec kernel's flush loops)
        // (on a stream of its own beside the entropy stage, whose serial chains leave the memory system idle: 4 GiB for a 1 GiB
        // frame, 1.2 ms that used to stand in front of it; joined again in front of the exec kernel)
        fill_st = (side && side->part_stream[2] && s->fill_fork && s->fill_join) ? side->part_stream[2] : st;
        if (fill_st != st && (hipEventRecord(s->fill_fork, st) != hipSuccess || hipStreamWaitEvent(fill_st, s->fill_fork, 0) != hipSuccess)) return ZL_ERR(ZL_E_GENERIC);
#ifdef ZL_DEV_SWITCHES
        if (!getenv("ZL_DEV_NO_FILL"))  // (experiment, round 6: what the fill costs the stages beside it -- this library's 1 GiB frame: entropy 3.33 -> 2.83 ms, exec 2.83 -> 2.59, the decode 10.2 -> 9.5 ms; the output is then wrong.
                                        // Doing without it means links written for every byte of every region that holds a waiting one, and a flag read per hop of the ordered copies: not built)
#endif
        if (!(ZL3_KEEP_LINKS && s->link_clean && span * 4 + 64 <= s->link_clean_bytes)) {
        if (hipMemsetAsync(s->link.p, 0xFF, span * 4 + 64, fill_st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        s->link_clean_bytes = std::max<size_t>(s->link_clean && ZL3_KEEP_LINKS ? s->link_clean_bytes : 0, (size_t)(span * 4 + 64));
        }
        s->link_clean = false;  // (until this decode has ended well)
        if (fill_st != st && hipEventRecord(s->fill_join, fill_st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        a.frames = d.frames; a.plan = d.plan; a.frec = d.frec; a.brec = d.brec; a.nframes = info.nsegs; a.src = d_src; a.dst = d_dst; a.results = nullptr;
        a.lit_scratch = (u8*)s->lit.p; a.seq_scratch = (u32*)s->seq.p; a.ignore_checksum = 1; a.stats = nullptr; a.max_blocks = d.max_blocks; a.order = (u32*)s->order.p;
        a.ext = d.ext; a.big = (const zl3_big*)s->big.p;
        a.ext_shared = 0; a.ext_base = nullptr; a.dict_ext_base = d.dict_ext ? hint->dict_ext_base : nullptr; a.dict_pre = nullptr; a.dict_end = nullptr; a.dict_len = 0; a.dict_id = 0; a.dict_rep[0] = 1; a.dict_rep[1] = 4; a.dict_rep[2] = 8;  // (the dictionary of a large frame travels in zl3_big)
        a.fuse1 = 3u | (ZL_QUAD_SPLIT ? 4u : 0u);
        a.part = 0;
        a.sym_first = (have_dict && hint->dict_formatted) ? 1u : 0u;
        stamp("alloc");
        if (zl2_launch_segments_entropy(&a, st, side) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stamp("entropy");
        }  // (phase != 2)
        if (phase == 1) {
            s->sv.d = d; s->sv.a = a; s->sv.info = info; s->sv.span = span; s->sv.fill_st = fill_st; s->sv.valid = true;
            return 0;
        }
        a.dst = d_dst;
        a.src = d_src;  // (a caller that receives the frame piece by piece may have moved it to a larger buffer since the first call)
        hipLaunchKernelGGL(zl3_totals_kernel, dim3((info.nblocks + 255) / 256), dim3(256), 0, st, d, (const u32*)s->seq.p, (uint4*)s->tot.p);
        {   // the history the frame starts with: a formatted dictionary's three offsets, else 1, 4, 8
            const bool dr = have_dict && hint->dict_formatted;
            hipLaunchKernelGGL(zl3_offsets_kernel, dim3(1), dim3(64 * ZL3_OFF_WAVES), 0, st, d, fd, (const uint4*)s->tot.p, dr ? hint->dict_rep[0] : 1u, dr ? hint->dict_rep[1] : 4u, dr ? hint->dict_rep[2] : 8u);
        }
        stamp("offsets");
        if (fill_st != st && hipStreamWaitEvent(st, s->fill_join, 0) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (zl2_launch_segments_exec(&a, exec_run, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stamp("exec");
        // the deferred bytes: one launch per group of output, in order (the first group has nothing in front of it to wait for,
        // but bytes of its later runs may wait for its earlier ones)
        const bool sink = hint && hint->sink_host && hint->sunk && info.fcs != ZL_CONTENTSIZE_UNKNOWN && span >= ((u64)2 << ZL3_SINK_LOG) && side && side->part_stream[1];
        u32 nsink = 0;
        // (this library's own 1 GiB frame, groups of 4 / 8 / 16 / 32 / 64 MiB: 12.6 / 12.8 / 13.6 / 14.7 / 16.6 ms -- its text quarter's
        // walks grow with the group as the reference's do)
        u32 glog = ZL3_GROUP_LOG;
#ifdef ZL_DEV_SWITCHES
        if (getenv("ZL_GROUP_LOG")) glog = (u32)atoi(getenv("ZL_GROUP_LOG"));
#endif
        // Groups without a single waiting byte -- raw blocks, blocks of literals only, matches that stay inside their exec run -- need
        // no launch: which ones those are comes down behind the exec kernel (one more wait for the stream; half of the 256 launches
        // of the 1 GiB mix, 9 - 16 us each, were for nothing).  Not with the host download beside this stage, whose events count launches.
        const u32 ngroups = (u32)((span + (((u64)1 << glog) - 1)) >> glog);
        const u32* gflags = nullptr;
        if (!sink && ngroups > 8 && s->gflag.ensure((size_t)ngroups * 4) && s->h_gflag.ensure((size_t)ngroups * 4)) {
            hipLaunchKernelGGL(zl3_group_flags_kernel, dim3(ngroups), dim3(256), 0, st, (const u32*)s->tile_flag.p, (u32*)s->gflag.p, (1u << glog) >> ZL3_TILE_LOG, ntiles);
            if (hipMemcpyAsync(s->h_gflag.p, s->gflag.p, (size_t)ngroups * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            gflags = (const u32*)s->h_gflag.p;
        }
        // Putting the links back costs the ordered launches a second look at every group's links (the 1 GiB reference frame: 19.5 -> 19.8 ms)
        // and saves the decode behind this one its fill -- which costs a frame of small blocks 0.6 - 0.9 ms (its literal and sequence lanes
        // read and write memory all the time: this library's own 1 GiB frame 10.2 -> 9.6 - 9.9 ms) and a frame of 128 KiB blocks next to
        // nothing (the fill runs beside chains of 22 000 sequences that leave the memory system idle).  So: frames whose segments hold
        // several blocks leave their links clean, the others leave them as they are and the next decode fills.
        const bool keep = ZL3_KEEP_LINKS && (u64)info.nsegs * 2 <= info.nblocks;
        u64 launched = ZL3_NO_RESET;  // the group of the launch in front: its links are put back by the next launch (zl3_group_copy_kernel)
        for (u64 g = 0; g < span; g += (u64)1 << glog) {
            if (gflags && !gflags[g >> glog]) continue;
            hipLaunchKernelGGL(zl3_group_copy_kernel, dim3((1u << glog) / ZL3_COPY_SLICE), dim3(ZL3_COPY_THREADS), 0, st,
                               d_dst + fd.dst_off, (u32*)s->link.p, (const u32*)s->tile_flag.p, g, keep ? launched : ZL3_NO_RESET, d.info);
            launched = g;
            const u64 done = g + ((u64)1 << glog);
            if (sink && (done & (((u64)1 << ZL3_SINK_LOG) - 1)) == 0 && done < span && nsink < ZL3_SINK_EVENTS) {
                if (!s->sink_ev[nsink] && hipEventCreateWithFlags(&s->sink_ev[nsink], hipEventDisableTiming) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
                if (hipEventRecord(s->sink_ev[nsink], st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
                nsink++;
            }
        }
        if (keep && launched != ZL3_NO_RESET)  // (the last group's links: a launch that copies nothing)
            hipLaunchKernelGGL(zl3_group_copy_kernel, dim3((1u << glog) / ZL3_COPY_SLICE), dim3(ZL3_COPY_THREADS), 0, st,
                               d_dst + fd.dst_off, (u32*)s->link.p, (const u32*)s->tile_flag.p, (u64)1 << 62, launched, d.info);
        // the download of what is finished, piece by piece behind the events, on a stream of its own (the host thread waits in
        // these calls when the memory is pageable; the copy kernels are all queued by now).  With a content checksum to check
        // (sink_hash) a second thread runs the XXH64 chain over every piece as soon as the event behind its copy has fired: the
        // calling thread is busy inside the copies, and hashing 1 GiB behind them instead of beside them costs 22 - 38 ms.
        std::atomic<u32> recorded{0};
        std::atomic<bool> hash_bad{false};
        std::thread hash_thread;
        if (nsink && hint->sink_hash) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            for (u32 k = 0; k < nsink; k++)
                if (!s->sink_done[k] && hipEventCreateWithFlags(&s->sink_done[k], hipEventDisableTiming) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            hash_thread = std::thread([&, dev]() {
                (void)hipSetDevice(dev);
                zlx::Xxh64* hx = (zlx::Xxh64*)hint->sink_hash;
                for (u32 k = 0; k < nsink; k++) {
                    u32 seen;
                    while ((seen = recorded.load(std::memory_order_acquire)) <= k && seen != 0xFFFFFFFFu) std::this_thread::sleep_for(std::chrono::microseconds(20));
                    if (seen == 0xFFFFFFFFu || hipEventSynchronize(s->sink_done[k]) != hipSuccess) { hash_bad.store(true); return; }
                    hx->update(hint->sink_host + fd.dst_off + ((u64)k << ZL3_SINK_LOG), (size_t)1 << ZL3_SINK_LOG);
                }
            });
        }
        bool copies_ok = true;
        for (u32 k = 0; k < nsink && copies_ok; k++) {
            const u64 off = (u64)k << ZL3_SINK_LOG;
            copies_ok = hipStreamWaitEvent(side->part_stream[1], s->sink_ev[k], 0) == hipSuccess &&
                        hipMemcpyAsync(hint->sink_host + fd.dst_off + off, d_dst + fd.dst_off + off, (size_t)1 << ZL3_SINK_LOG, hipMemcpyDeviceToHost, side->part_stream[1]) == hipSuccess;
            if (copies_ok && hash_thread.joinable()) {
                copies_ok = hipEventRecord(s->sink_done[k], side->part_stream[1]) == hipSuccess;
                if (copies_ok) recorded.store(k + 1, std::memory_order_release);
            }
        }
        if (hash_thread.joinable()) {
            if (!copies_ok) recorded.store(0xFFFFFFFFu, std::memory_order_release);
            hash_thread.join();
        }
        if (!copies_ok || hash_bad.load()) return ZL_ERR(ZL_E_GENERIC);
        if (nsink && hipStreamSynchronize(side->part_stream[1]) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (hipMemcpyAsync(&info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stamp("resolve");
        if (info.err == ZL_E_FALLBACK) return ZL_ERR(ZL_E_FALLBACK);
        s->link_clean = keep && !info.err;  // (every launch above has run: the stream was drained for `info`)
        if (nsink && !info.err) *hint->sunk = (u64)nsink << ZL3_SINK_LOG;  // (a frame that failed or goes to the general kernel is downloaded as a whole, whatever was copied)
        if (hint && hint->run_out && !info.err) { hint->run_out[0] = info.produced; for (int k = 0; k < 3; k++) hint->run_out[1 + k] = info.rep_end[k]; }
    }
    if (phase == 1) {  // (the walk found the run damaged: phase 2 writes that verdict)
        s->sv.d = d; s->sv.a = a; s->sv.info = info; s->sv.span = span; s->sv.fill_st = fill_st; s->sv.valid = true;
        return 0;
    }
    u32 have_hash = 0, hash = 0;
    if (!ignore_checksum && !info.err && info.has_checksum) {  // (the host entries come with ignore_checksum set: they hash what they have downloaded)
        if (!zl3_hash_on_host(s, d_dst + fd.dst_off, info.produced, st, &hash)) return ZL_ERR(ZL_E_GENERIC);
        have_hash = 1;
    }
    hipLaunchKernelGGL(zl3_finish_kernel, dim3(1), dim3(512), 0, st, d.info, fd, d_src, (const u8*)d_dst, d_result, (u32)ignore_checksum, have_hash, hash);
    stamp("finish");
    return hipGetLastError() == hipSuccess ? 0 : ZL_ERR(ZL_E_GENERIC);
}
