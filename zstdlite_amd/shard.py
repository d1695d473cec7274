"""Multi-GPU sharding of the hot path.

zstd frames are self-contained, so a large raw vector is cut into 128 KiB chunks (a pure
function of its size, never of the GPU count, which keeps the output deterministic), rank g
owns one contiguous range of chunks, and compress / decompress need no data-path collective.
The one exchange step is collecting the compressed shards: an all-gather of the shard sizes
(8 bytes per rank) turns them into offsets, then every rank sends its payload straight to the
root over its own xGMI link (grouped send/recv; RCCL when the backend is "nccl", gloo in the
CPU tests).  The reference has no counterpart: its only parallelism is libzstd's pthread job
pool (zstd.c:36165-38048), whose 8 MiB jobs with overlap are the precedent for chunking.
"""
import torch
import torch.distributed as dist

CHUNK = 128 << 10


def plan_shards(nbytes, world):
    """Contiguous byte ranges, aligned to the chunk size, one per rank: [(start, stop)]."""
    nchunks = max(1, -(-nbytes // CHUNK))
    base, extra = divmod(nchunks, world)
    out, c = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append((min(c * CHUNK, nbytes), min((c + k) * CHUNK, nbytes)))
        c += k
    return out


def gather_compressed(payload, total, root=0, group=None, out=None, wait=True):
    """Collect every rank's compressed shard (payload[:total], uint8 tensor) on `root`, in rank
    order.  Returns (buffer, sizes) on root and (None, sizes) elsewhere.  `out` may be a
    preallocated root buffer that is reused when large enough.  With wait=False the transfers are
    only started: a third return value holds the work handles, and the caller overlaps them with
    compute (the payload must stay untouched until they have been waited for)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    # P2POp peers are GLOBAL ranks; inside a sub-group the group-relative numbers must be translated
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([int(total)], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    hs = [int(x) for x in sizes.cpu().tolist()]
    if rank == root:
        need = sum(hs)
        if out is None or out.numel() < need:
            out = torch.empty(need + (need >> 4) + 1024, dtype=torch.uint8, device=dev)
        ops, off = [], 0
        for r in range(world):
            if r == root:
                out[off:off + hs[r]].copy_(payload[:hs[r]])
            elif hs[r]:
                ops.append(dist.P2POp(dist.irecv, out[off:off + hs[r]], peer(r), group=group))
            off += hs[r]
    else:
        ops = [dist.P2POp(dist.isend, payload[:int(total)], peer(root), group=group)] if total else []
    works = dist.batch_isend_irecv(ops) if ops else []
    if not wait:
        return (out if rank == root else None), hs, works
    for w in works:
        w.wait()
    return (out if rank == root else None), hs
