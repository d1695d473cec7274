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
//   zl3_parse_kernel    lane per block: literal / sequence section headers               zstd.c:43146, :43707
//   zl3_plan_kernel     the frame becomes SEGMENTS of ZL3_SEG_BLOCKS blocks; for each, the earlier blocks whose
//                       Huffman tree / FSE tables it inherits are found with a running maximum
//   zl2_lit_kernel, zl2_seq_kernel (zl_decode2.hip)  the segments go through the lane pipeline's entropy
//                       kernels as if they were frames; inherited tables are rebuilt in extra rounds;
//                       repcodes stay symbolic for every block but the first
//   zl3_offsets_kernel  one wave: output position and starting repcode history of every block
//   zl2_exec_kernel<true, false>  workgroup per run of 8 segments, all runs at once: bytes whose source lies in
//                       another segment are left as links (frame positions)
//   zl3_copy_kernel     the deferred bytes, in order: one tiny launch per run of segments, so that every
//                       source is final when it is read (zl3_resolve_kernel: the pointer-jumping variant)
//   zl3_finish_kernel   content size, XXH64 (zstd.c:41646-41660), the frame's result word
//
// All integer work; output must be bit-exact with the reference decoder.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "zl_blocks.h"
#include "zl_decode3.h"
#include "zl_device.h"
#include "zl_format.h"
#include "zl_host.h"
#include "zl_kernels.h"

namespace {

struct zl3_dev {  // device pointers of one large-frame decode
    zl3_info* info;
    zl2_blk_rec* brec;
    zl2_frame_rec* frec;
    zl2_frame_plan* plan;
    zl_frame_desc* frames;
    zl2_blk_rec* ext;
    u32* blk_dst;
    u32* rep_start;
    u32* seg_tmp;   // per segment 16 words (zl3_plan_kernel): [0..5] its own numbers, [6..13] the scanned ones
    u32* max_blocks;
    u32 cap_blocks;
};

// record of block b of the frame: segment b / ZL3_SEG_BLOCKS, slot b % ZL3_SEG_BLOCKS of its ZL2_MAXB records
__device__ __forceinline__ size_t rec_of(u32 b) { return (size_t)(b / ZL3_SEG_BLOCKS) * ZL2_MAXB + b % ZL3_SEG_BLOCKS; }
__device__ __forceinline__ void info_err(zl3_info* info, u32 code) { atomicCAS(&info->err, 0u, code); }

// ---------------------------------------------------------------------------------------
// frame header + block walk: one lane (zstd.c:41050-41148, :43075)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void zl3_walk_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base) {
    if (threadIdx.x != 0) return;
    zl3_info* info = d.info;
    const u8* src = src_base + fd.src_off;
    const u64 n = fd.src_size;
    if (n < 5) { info->err = ZL_E_SRCSIZE_WRONG; return; }
    if (zl_rd_le(src, 4) != ZL_MAGIC) { info->err = ZL_E_PREFIX_UNKNOWN; return; }
    const u32 fhd = src[4];
    const u32 dflag = fhd & 3, single = (fhd >> 5) & 1, fflag = fhd >> 6;
    if (fhd & 8) { info->err = ZL_E_FRAMEPARAM_UNSUPPORTED; return; }
    const u32 dbytes = dflag == 3 ? 4 : dflag;
    const u32 fbytes = fflag == 0 ? (single ? 1 : 0) : (1u << fflag);
    const u32 hs = 5 + (single ? 0 : 1) + dbytes + fbytes;
    if (n < hs) { info->err = ZL_E_SRCSIZE_WRONG; return; }
    u32 pos = 5;
    if (!single) {
        const u32 wlog = (src[pos++] >> 3) + 10;
        if (wlog > 31) { info->err = ZL_E_WINDOW_TOO_LARGE; return; }
    }
    const u32 did = zl_rd_le(src + pos, dbytes);
    pos += dbytes;
    u64 fcs = ZL_CONTENTSIZE_UNKNOWN;
    if (fflag == 0) { if (single) fcs = src[pos]; }
    else if (fflag == 1) fcs = (u64)zl_rd_le(src + pos, 2) + 256;
    else if (fflag == 2) fcs = zl_rd_le(src + pos, 4);
    else fcs = (u64)zl_rd_le(src + pos, 4) | ((u64)zl_rd_le(src + pos + 4, 4) << 32);
    if (did) { info->err = ZL_E_DICT_WRONG; return; }  // this path never has a dictionary (zstd.c:41318)
    // (a frame that does not record its size -- what ZSTD_compressStream2 writes when nothing is pledged -- is
    // bounded by its block count: the host sizes the links from that)
    if (fcs != ZL_CONTENTSIZE_UNKNOWN && fcs > fd.dst_cap) { info->err = ZL_E_DSTSIZE_TOO_SMALL; return; }
    info->fcs = fcs;
    info->has_checksum = (fhd >> 2) & 1;
    pos = hs;
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
        zl2_blk_rec* r = d.brec + rec_of(nb);
        r->type = type; r->size = bsize; r->src_off = pos;
        nb++;
        pos += csize;
        if (last) break;
    }
    if (info->has_checksum && n - pos < 4) { info->err = ZL_E_CHECKSUM_WRONG; return; }
    info->nblocks = nb;
    info->nsegs = (nb + ZL3_SEG_BLOCKS - 1) / ZL3_SEG_BLOCKS;
    info->end_pos = pos;
}

// ---------------------------------------------------------------------------------------
// lane per block: section headers
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zl3_parse_kernel(zl3_dev d, zl_frame_desc fd, const u8* src_base) {
    const u32 b = blockIdx.x * 256 + threadIdx.x;
    zl3_info* info = d.info;
    if (info->err || b >= info->nblocks) return;
    zl2_blk_rec r = d.brec[rec_of(b)];
    r.lit_mode = 0; r.lit_size = 0; r.lit_a = 0; r.lit_b = 0; r.seq_off = 0; r.seq_len = 0;
    r.lit_dst = 0; r.seq_dst = 0; r.nseq = 0; r.streams = 0; r.seq_modes = 0; r.pad[0] = 0; r.pad[1] = 0;
    if (r.type == 2) {
        u32 lit_bytes = 0;
        const u32 e = parse_block_sections(src_base + fd.src_off + r.src_off, r.size, r, lit_bytes);
        if (e) { info_err(info, e); return; }
        r.pad[0] = lit_bytes;  // (scratch the block's Huffman literals need; consumed by zl3_plan_kernel)
    }
    d.brec[rec_of(b)] = r;
}

// ---------------------------------------------------------------------------------------
// segments: scratch offsets and inherited tables.  One workgroup.
// ---------------------------------------------------------------------------------------
#define ZL3_NONE 0xFFFFFFFFu
__device__ __forceinline__ u32 max_def(u32 a, u32 b) { return (a == ZL3_NONE) ? b : ((b == ZL3_NONE) ? a : (a > b ? a : b)); }

__global__ __launch_bounds__(1024) void zl3_plan_kernel(zl3_dev d, zl_frame_desc fd) {
    __shared__ u32 tile[6][1024];
    __shared__ u64 carry_sum[2];
    __shared__ u32 carry_def[4];
    zl3_info* info = d.info;
    if (info->err) return;
    const u32 t = threadIdx.x, nb = info->nblocks, ns = info->nsegs;
    // ---- A: inside each segment: where its blocks' literals and sequences go, totals, last defining blocks
    for (u32 s = t; s < ns; s += 1024) {
        u32 lit = 0, seq = 0, def[4] = {ZL3_NONE, ZL3_NONE, ZL3_NONE, ZL3_NONE};
        for (u32 k = 0; k < ZL3_SEG_BLOCKS; k++) {
            const u32 b = s * ZL3_SEG_BLOCKS + k;
            if (b >= nb) break;
            zl2_blk_rec* r = d.brec + rec_of(b);
            if (r->type != 2) continue;
            r->lit_dst = lit;
            r->seq_dst = seq;
            lit += r->pad[0];
            seq += r->nseq + 2;  // + the totals and final-repcode records the sequence lanes append
            if (r->lit_mode == 2) def[0] = b;
            if (r->nseq > 0 && r->seq_modes != 0x100u) {
                const u32 m = r->seq_modes;
                if ((m >> 6) != 3) def[1] = b;
                if (((m >> 4) & 3) != 3) def[2] = b;
                if (((m >> 2) & 3) != 3) def[3] = b;
            }
        }
        u32* o = d.seg_tmp + (size_t)s * 16;
        o[0] = (lit + 15) & ~15u; o[1] = seq; o[2] = def[0]; o[3] = def[1]; o[4] = def[2]; o[5] = def[3];
    }
    if (t == 0) { carry_sum[0] = 0; carry_sum[1] = 0; carry_def[0] = carry_def[1] = carry_def[2] = carry_def[3] = ZL3_NONE; }
    zl_sync_global();  // (the segments' numbers travel through memory)
    // ---- B: across segments: exclusive sums of the scratch needs, exclusive running maximum of the defining blocks
    for (u32 s0 = 0; s0 < ns; s0 += 1024) {
        const u32 cnt = ns - s0 < 1024 ? ns - s0 : 1024;
        if (t < cnt) for (int q = 0; q < 6; q++) tile[q][t] = d.seg_tmp[(size_t)(s0 + t) * 16 + q];
        __syncthreads();
        if (t < 6) {  // one lane per column
            if (t < 2) {
                u64 run = carry_sum[t];
                for (u32 i = 0; i < cnt; i++) {
                    const u32 v = tile[t][i];
                    d.seg_tmp[(size_t)(s0 + i) * 16 + 6 + t] = (u32)run;  // (64-bit offsets: low word, high word)
                    d.seg_tmp[(size_t)(s0 + i) * 16 + 12 + t] = (u32)(run >> 32);
                    run += v;
                }
                carry_sum[t] = run;
            } else {
                u32 run = carry_def[t - 2];
                for (u32 i = 0; i < cnt; i++) {
                    const u32 v = tile[t][i];
                    d.seg_tmp[(size_t)(s0 + i) * 16 + 6 + t] = run;  // words 8..11
                    run = max_def(run, v);
                }
                carry_def[t - 2] = run;
            }
        }
        __syncthreads();
    }
    zl_sync_global();
    // ---- C: segment records
    for (u32 s = t; s < ns; s += 1024) {
        const u32* o = d.seg_tmp + (size_t)s * 16;
        zl2_frame_plan p;
        p.lit_off = (u64)o[6] | ((u64)o[12] << 32);
        p.seq_off = (u64)o[7] | ((u64)o[13] << 32);
        p.lit_cap = o[0]; p.seq_cap = o[1]; p.route = 1; p.pad = 0;
        d.plan[s] = p;
        zl2_frame_rec fr;
        const u32 left = nb - s * ZL3_SEG_BLOCKS;
        fr.err = 0; fr.nblocks = left < ZL3_SEG_BLOCKS ? left : ZL3_SEG_BLOCKS; fr.has_checksum = 0; fr.skippable = 0;
        fr.fcs = ZL_CONTENTSIZE_UNKNOWN; fr.end_pos = 0; fr.cost = o[1]; fr.lit_cost = o[0]; fr.pad2 = 0;
        d.frec[s] = fr;
        d.frames[s] = fd;
        // inherited tables; every repeat inside the segment must have something to repeat (zstd.c:43171, :43666)
        u32 cur[4] = {o[8], o[9], o[10], o[11]};  // latest defining block in front of the segment: Huffman tree, LL, OF, ML
        zl2_blk_rec none;
        none.type = ZL3_NONE;
        zl2_blk_rec* x = d.ext + (size_t)s * 4;
        x[0] = none; x[1] = none; x[2] = none; x[3] = none;
        if (cur[0] != ZL3_NONE) x[0] = d.brec[rec_of(cur[0])];
        {   // the (up to three) distinct defining blocks of the sequence tables, in block order
            u32 a0 = cur[1], a1 = cur[2], a2 = cur[3];
            if (a0 > a1) { const u32 q = a0; a0 = a1; a1 = q; }   // (ZL3_NONE sorts last)
            if (a1 > a2) { const u32 q = a1; a1 = a2; a2 = q; }
            if (a0 > a1) { const u32 q = a0; a0 = a1; a1 = q; }
            u32 k = 1;
            if (a0 != ZL3_NONE) x[k++] = d.brec[rec_of(a0)];
            if (a1 != ZL3_NONE && a1 != a0) x[k++] = d.brec[rec_of(a1)];
            if (a2 != ZL3_NONE && a2 != a1) x[k++] = d.brec[rec_of(a2)];
        }
        for (u32 k = 0; k < fr.nblocks; k++) {
            const u32 b = s * ZL3_SEG_BLOCKS + k;
            const zl2_blk_rec* r = d.brec + rec_of(b);
            if (r->type != 2) continue;
            if (r->lit_mode == 3 && cur[0] == ZL3_NONE) info_err(info, ZL_E_DICT_CORRUPTED);
            if (r->lit_mode == 2) cur[0] = b;
            if (r->nseq > 0 && r->seq_modes != 0x100u) {
                const u32 m = r->seq_modes;
                const u32 mk[3] = {m >> 6, (m >> 4) & 3, (m >> 2) & 3};
                for (int q = 0; q < 3; q++) {
                    if (mk[q] == 3) { if (cur[1 + q] == ZL3_NONE) info_err(info, ZL_E_CORRUPTION); }
                    else cur[1 + q] = b;
                }
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        info->lit_total = carry_sum[0];
        info->seq_total = carry_sum[1];
        *d.max_blocks = ZL3_SEG_BLOCKS;  // (which literal-kernel instantiation the segments take)
    }
}

// ---------------------------------------------------------------------------------------
// after the entropy kernels: output position and starting repcode history of every block.  One wave;
// the per-block numbers are fetched 64 blocks at a time, the chain itself runs on lane 0 out of LDS.
// ---------------------------------------------------------------------------------------
#define ZL3_RESOLVE(v_, r0_, r1_, r2_) (((v_) >> 31) ? (((v_) & 3u) == 0 ? (r0_) : (((v_) & 3u) == 1 ? (r1_) : (r2_))) - (((v_) & 0x7FFFFFFFu) >> 2) : (v_))
__global__ __launch_bounds__(64) void zl3_offsets_kernel(zl3_dev d, zl_frame_desc fd, const u32* seq_scratch) {
    __shared__ u32 outsz[64], fin[64][3], has[64], o_dst[64], o_rep[64][3];
    zl3_info* info = d.info;
    const u32 lane = threadIdx.x, nb = info->nblocks, ns = info->nsegs;
    if (info->err) return;
    // the segments' errors first (lowest segment wins, as the reference would meet it first)
    {
        u32 first = ZL3_NONE;
        for (u32 s = lane; s < ns; s += 64) if (d.frec[s].err && s < first) first = s;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) { const u32 o = __shfl_xor(first, k, 64); first = o < first ? o : first; }
        if (first != ZL3_NONE) { if (lane == 0) info->err = d.frec[first].err; return; }
    }
    u64 op = 0;
    u32 r0 = 1, r1 = 4, r2 = 8;  // zstd.c:41433
    u32 bad = 0;
    for (u32 b0 = 0; b0 < nb; b0 += 64) {
        const u32 b = b0 + lane;
        u32 sz = 0, h = 0, f0 = 0, f1 = 0, f2 = 0;
        if (b < nb) {
            const zl2_blk_rec r = d.brec[rec_of(b)];
            if (r.type != 2) sz = r.size;
            else if (r.nseq == 0) sz = r.lit_size;
            else {
                const uint4* q = (const uint4*)seq_scratch + d.plan[b / ZL3_SEG_BLOCKS].seq_off + r.seq_dst;
                const uint4 tot = q[r.nseq], fh = q[r.nseq + 1];
                sz = tot.y + (r.lit_size - tot.x);
                h = 1; f0 = fh.x; f1 = fh.y; f2 = fh.z;
            }
        }
        outsz[lane] = sz; has[lane] = h; fin[lane][0] = f0; fin[lane][1] = f1; fin[lane][2] = f2;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane == 0) {
            const u32 cnt = nb - b0 < 64 ? nb - b0 : 64;
            for (u32 i = 0; i < cnt; i++) {
                o_dst[i] = (u32)op;
                o_rep[i][0] = r0; o_rep[i][1] = r1; o_rep[i][2] = r2;
                if (outsz[i] > ZL_BLOCK_MAX) bad = ZL_E_CORRUPTION;
                op += outsz[i];
                if (has[i]) {
                    const u32 n0 = ZL3_RESOLVE(fin[i][0], r0, r1, r2), n1 = ZL3_RESOLVE(fin[i][1], r0, r1, r2), n2 = ZL3_RESOLVE(fin[i][2], r0, r1, r2);
                    r0 = n0; r1 = n1; r2 = n2;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (b < nb) {
            d.blk_dst[b] = o_dst[lane];
            d.rep_start[(size_t)b * 3 + 0] = o_rep[lane][0];
            d.rep_start[(size_t)b * 3 + 1] = o_rep[lane][1];
            d.rep_start[(size_t)b * 3 + 2] = o_rep[lane][2];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        d.blk_dst[nb] = (u32)op;
        info->produced = op;
        if (bad) info->err = bad;
        else if (op > fd.dst_cap) info->err = ZL_E_DSTSIZE_TOO_SMALL;
        else if (info->fcs != ZL_CONTENTSIZE_UNKNOWN && op != info->fcs) info->err = ZL_E_CORRUPTION;  // zstd.c:41646
    }
}

// ---------------------------------------------------------------------------------------
// resolve rounds: pointer jumping over the deferred bytes.  A byte whose source became final in an
// EARLIER launch copies it and becomes final itself (marked with this round's value, so that bytes
// that became final during this launch are not read before the launch has ended); otherwise it takes
// over its source's link.  Tiles without pending bytes are skipped.
// ---------------------------------------------------------------------------------------
#define ZL3_MARK_MIN (ZL3_ROOT - 2)
__global__ __launch_bounds__(256) void zl3_resolve_kernel(u8* dst, u32* link, u32* tile_flag, u64 n, u32 round, zl3_info* info) {
    __shared__ u32 wg_pending;
    const u32 tile = blockIdx.x, t = threadIdx.x;
    if (!tile_flag[tile]) return;
    if (t == 0) wg_pending = 0;
    __syncthreads();
    const u32 cur = (round & 1) ? ZL3_ROOT - 1 : ZL3_ROOT - 2;
    const u64 base = (u64)tile << ZL3_TILE_LOG;
    u32 pending = 0;
    for (u32 i0 = t * 4; i0 < (1u << ZL3_TILE_LOG); i0 += 256 * 4) {
        const u64 p0 = base + i0;
        if (p0 >= n) break;
        u32 lk[4];
        if (p0 + 4 <= n) { const uint4 v = *(const uint4*)(link + p0); lk[0] = v.x; lk[1] = v.y; lk[2] = v.z; lk[3] = v.w; }
        else for (int j = 0; j < 4; j++) lk[j] = (p0 + j < n) ? link[p0 + j] : ZL3_ROOT;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 l1 = lk[j];
            if (l1 >= ZL3_MARK_MIN) continue;
            if ((u64)l1 >= p0 + j) { info_err(info, ZL_E_CORRUPTION); continue; }  // (links only ever point backwards; a guard against garbage)
            const u32 l2 = link[l1];
            if (l2 >= ZL3_MARK_MIN) {
                if (l2 != cur) { dst[p0 + j] = dst[l1]; link[p0 + j] = cur; }
                else pending++;
            } else {
                link[p0 + j] = l2;
                pending++;
            }
        }
    }
    if (pending) atomicAdd(&wg_pending, pending);
    __syncthreads();
    if (t == 0) {
        if (wg_pending) atomicAdd(&info->pending, wg_pending);
        else tile_flag[tile] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// resolve, in order: one launch per run of segments (the unit an exec workgroup worked through).  A
// deferred byte of run R copies a byte of an earlier run, and all of those are final once the
// launches of the runs in front of R have ended -- so every deferred byte is visited exactly once and
// no link is ever followed twice.  The launches are tiny (<= 1 MiB of output each); what they cost is
// their number, not their work.
// ---------------------------------------------------------------------------------------
#define ZL3_COPY_THREADS 256
#define ZL3_COPY_SLICE 4096u  // output bytes per workgroup
__global__ __launch_bounds__(ZL3_COPY_THREADS) void zl3_copy_kernel(u8* dst, const u32* link, const u32* tile_flag, const u32* blk_dst, u32 run, zl3_info* info) {
    const u32 nb = info->nblocks;
    const u32 b0 = run * ZL3_EXEC_RUN * ZL3_SEG_BLOCKS, b1 = b0 + ZL3_EXEC_RUN * ZL3_SEG_BLOCKS < nb ? b0 + ZL3_EXEC_RUN * ZL3_SEG_BLOCKS : nb;
    const u32 p0 = blk_dst[b0], p1 = blk_dst[b1];
    const u32 s0 = (p0 & ~3u) + blockIdx.x * ZL3_COPY_SLICE;
    if (s0 >= p1) return;
    for (u32 q = s0 + threadIdx.x * 4; q < s0 + ZL3_COPY_SLICE && q < p1; q += ZL3_COPY_THREADS * 4) {
        if (!tile_flag[q >> ZL3_TILE_LOG]) continue;
        const uint4 v = *(const uint4*)(link + q);  // (the link buffer is padded: reading a few entries past the end is fine)
        const u32 lk[4] = {v.x, v.y, v.z, v.w};
        u32 word = 0;
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 pos = q + j;
            if (pos < p0 || pos >= p1 || lk[j] >= ZL3_MARK_MIN) continue;
            if (!any) { __builtin_memcpy(&word, dst + q, 4); any = true; }  // (the frame may start at any address)
            if (lk[j] >= pos) { info_err(info, ZL_E_CORRUPTION); continue; }  // (links only ever point backwards; a guard against garbage)
            word = (word & ~(0xFFu << (8 * j))) | ((u32)dst[lk[j]] << (8 * j));
        }
        if (any) __builtin_memcpy(dst + q, &word, 4);
    }
}

// ---------------------------------------------------------------------------------------
// the frame's result: checksum and size (zstd.c:41646-41660)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void zl3_finish_kernel(zl3_info* info, zl_frame_desc fd, const u8* src_base, const u8* dst_base, u64* result, u32 ignore_checksum) {
    __shared__ __align__(16) u8 tile[8192];
    __shared__ u64 acc[4];
    __shared__ u32 bc[4];
    u32 err = info->err;
    if (!err && info->pending) err = ZL_E_CORRUPTION;  // (cannot happen for links that only point backwards; kept as a guard)
    if (!err && info->has_checksum && !ignore_checksum) {
        const u32 got = zl_xxh64_low32(tile, acc, bc, dst_base + fd.dst_off, info->produced);
        if (got != zl_rd_le(src_base + fd.src_off + info->end_pos, 4)) err = ZL_E_CHECKSUM_WRONG;
    }
    if (threadIdx.x == 0) *result = err ? ZL_ERR(err) : info->produced;
}

}  // namespace

struct zl3_state_s {
    DevBuf info, brec, frec, plan, frames, ext, blk_dst, rep_start, seg_tmp, order, word, lit, seq, link, tile_flag, big;
};

extern "C" zl3_state* zl3_create(void) { return new (std::nothrow) zl3_state_s(); }
extern "C" void zl3_free(zl3_state* s) {
    if (!s) return;
    for (DevBuf* b : {&s->info, &s->brec, &s->frec, &s->plan, &s->frames, &s->ext, &s->blk_dst, &s->rep_start, &s->seg_tmp, &s->order, &s->word, &s->lit, &s->seq, &s->link, &s->tile_flag, &s->big}) b->release();
    delete s;
}

// Decodes one large frame.  Returns 0 (the result word of the frame has been written on the device),
// ZL_ERR(ZL_E_FALLBACK) when the frame has to take the general kernel, or another error of the launch itself.
extern "C" size_t zl3_decode_frame(zl3_state* s, const zl_frame_desc* fdp, const u8* d_src, u8* d_dst, u64* d_result, int ignore_checksum, hipStream_t st, const zl2_side* side) {
    const zl_frame_desc fd = *fdp;
    // ZL_DEBUG_BIG=1 in the environment prints the time of every stage to stderr (each stamp synchronises the stream)
    static const bool dbg = getenv("ZL_DEBUG_BIG") != nullptr;
    auto now = [&]() { (void)hipStreamSynchronize(st); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = dbg ? now() : 0;
    auto stamp = [&](const char* what) { if (dbg) { const double t1 = now(); fprintf(stderr, "[zl3] %-10s %8.3f ms\n", what, t1 - t0); t0 = t1; } };
    if (fd.src_size >= 0xFFFFFF00ull) return ZL_ERR(ZL_E_FALLBACK);  // block records are 32-bit
    // (dst_cap may be the room left in the caller's buffer rather than the frame's size: what the frame can produce is bounded by its input too)
    const u64 out_bound = std::min<u64>(fd.dst_cap, fd.src_size * 1024 + ((u64)1 << 20));
    // blocks planned for: full blocks plus a generous allowance for encoders that split blocks
    const u32 cap_blocks = (u32)std::min<u64>((out_bound >> 14) + (fd.src_size >> 11) + 64, 1u << 22) & ~(u32)(ZL2_MAXB - 1);
    const u32 cap_segs = cap_blocks / ZL3_SEG_BLOCKS;
    if (!s->info.ensure(sizeof(zl3_info)) || !s->brec.ensure((size_t)cap_segs * ZL2_MAXB * sizeof(zl2_blk_rec)) || !s->frec.ensure((size_t)cap_segs * sizeof(zl2_frame_rec)) ||
        !s->plan.ensure((size_t)cap_segs * sizeof(zl2_frame_plan)) || !s->frames.ensure((size_t)cap_segs * sizeof(zl_frame_desc)) ||
        !s->ext.ensure((size_t)cap_segs * 4 * sizeof(zl2_blk_rec)) || !s->blk_dst.ensure(((size_t)cap_blocks + 1) * 4) || !s->rep_start.ensure((size_t)cap_blocks * 12) ||
        !s->seg_tmp.ensure((size_t)cap_segs * 64) || !s->order.ensure((size_t)cap_segs * 8 + 64) || !s->word.ensure(64) || !s->big.ensure(sizeof(zl3_big)))
        return kGpuUnavailable;
    zl3_dev d;
    d.info = (zl3_info*)s->info.p; d.brec = (zl2_blk_rec*)s->brec.p; d.frec = (zl2_frame_rec*)s->frec.p; d.plan = (zl2_frame_plan*)s->plan.p;
    d.frames = (zl_frame_desc*)s->frames.p; d.ext = (zl2_blk_rec*)s->ext.p; d.blk_dst = (u32*)s->blk_dst.p; d.rep_start = (u32*)s->rep_start.p;
    d.seg_tmp = (u32*)s->seg_tmp.p; d.max_blocks = (u32*)s->word.p; d.cap_blocks = cap_blocks;
    if (hipMemsetAsync(d.info, 0, sizeof(zl3_info), st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    hipLaunchKernelGGL(zl3_walk_kernel, dim3(1), dim3(64), 0, st, d, fd, d_src);
    hipLaunchKernelGGL(zl3_parse_kernel, dim3((cap_blocks + 255) / 256), dim3(256), 0, st, d, fd, d_src);
    hipLaunchKernelGGL(zl3_plan_kernel, dim3(1), dim3(1024), 0, st, d, fd);
    zl3_info info;
    if (hipMemcpyAsync(&info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
    stamp("index");
    if (dbg) fprintf(stderr, "[zl3] blocks %u segments %u literals %llu sequences %llu err %u\n", info.nblocks, info.nsegs, (unsigned long long)info.lit_total, (unsigned long long)info.seq_total, info.err);
    if (info.err == ZL_E_FALLBACK) return ZL_ERR(ZL_E_FALLBACK);
    // output positions the links must cover: the recorded size, else what the blocks can produce at most (and the caller has room for)
    const u64 span = info.fcs != ZL_CONTENTSIZE_UNKNOWN ? info.fcs : std::min<u64>(fd.dst_cap, (u64)info.nblocks * ZL_BLOCK_MAX);
    if (span >= 0xFFFFFF00ull) return ZL_ERR(ZL_E_FALLBACK);  // links are 32-bit
    const u64 ntiles = (span >> ZL3_TILE_LOG) + 1;
    if (!info.err) {
        if (!s->lit.ensure(info.lit_total + 64) || !s->seq.ensure(info.seq_total * 16 + 64) || !s->link.ensure(span * 4 + 64) || !s->tile_flag.ensure(ntiles * 4))
            return kGpuUnavailable;
        zl3_big big;
        big.info = d.info; big.link = (u32*)s->link.p; big.tile_flag = (u32*)s->tile_flag.p; big.blk_dst = d.blk_dst; big.rep_start = d.rep_start;
        if (hipMemcpyAsync(s->big.p, &big, sizeof(big), hipMemcpyHostToDevice, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        if (hipMemsetAsync(s->tile_flag.p, 0, ntiles * 4, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        zl2_args a;
        a.frames = d.frames; a.plan = d.plan; a.frec = d.frec; a.brec = d.brec; a.nframes = info.nsegs; a.src = d_src; a.dst = d_dst; a.results = nullptr;
        a.lit_scratch = (u8*)s->lit.p; a.seq_scratch = (u32*)s->seq.p; a.ignore_checksum = 1; a.stats = nullptr; a.max_blocks = d.max_blocks; a.order = (u32*)s->order.p;
        a.ext = d.ext; a.big = (const zl3_big*)s->big.p;
        static const u32 fuse1 = getenv("ZL_NO_FUSE1") ? 0u : (getenv("ZL_NO_QUAD") ? 1u : 3u);
        a.fuse1 = fuse1;
        stamp("alloc");
        if (zl2_launch_segments_entropy(&a, st, side) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stamp("entropy");
        hipLaunchKernelGGL(zl3_offsets_kernel, dim3(1), dim3(64), 0, st, d, fd, (const u32*)s->seq.p);
        stamp("offsets");
        if (zl2_launch_segments_exec(&a, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        stamp("exec");
        if (hipMemcpyAsync(&info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
        static const bool jumping = getenv("ZL_BIG_JUMPING") != nullptr;  // (experiments: the pointer-jumping rounds instead of the in-order launches)
        if (!info.err && !jumping) {
            const u32 nruns = (info.nsegs + ZL3_EXEC_RUN - 1) / ZL3_EXEC_RUN;
            for (u32 run = 1; run < nruns; run++)
                hipLaunchKernelGGL(zl3_copy_kernel, dim3((ZL3_EXEC_RUN * ZL3_SEG_BLOCKS * ZL_BLOCK_MAX + 4) / ZL3_COPY_SLICE + 1), dim3(ZL3_COPY_THREADS), 0, st,
                                   d_dst + fd.dst_off, (const u32*)s->link.p, (const u32*)s->tile_flag.p, (const u32*)d.blk_dst, run, d.info);
        }
        for (u32 round = 0; !info.err && jumping; round++) {
            if (hipMemsetAsync(&d.info->pending, 0, 4, st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            hipLaunchKernelGGL(zl3_resolve_kernel, dim3((u32)ntiles), dim3(256), 0, st, d_dst + fd.dst_off, (u32*)s->link.p, (u32*)s->tile_flag.p, span, round, d.info);
            if (hipMemcpyAsync(&info, d.info, sizeof(info), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ZL_ERR(ZL_E_GENERIC);
            if (dbg) fprintf(stderr, "[zl3] round %u pending %u\n", round, info.pending);
            if (info.err || info.pending == 0 || round >= 80) break;
        }
        stamp("resolve");
        if (info.err == ZL_E_FALLBACK) return ZL_ERR(ZL_E_FALLBACK);
    }
    hipLaunchKernelGGL(zl3_finish_kernel, dim3(1), dim3(512), 0, st, d.info, fd, d_src, (const u8*)d_dst, d_result, (u32)ignore_checksum);
    stamp("finish");
    return hipGetLastError() == hipSuccess ? 0 : ZL_ERR(ZL_E_GENERIC);
}
