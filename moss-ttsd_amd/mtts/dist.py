"""Batch sharding over the GPUs of one node: dialogues are independent, so the
JSONL batch is dealt to ranks, every rank runs the full engine, and the only exchanges are a
weight broadcast at start-up and a length/audio gather at the end (RCCL over xGMI on GPUs, gloo
in the CPU tests).  No per-step collective, no tensor parallelism.

Shaped for xGMI (a full mesh of point-to-point links, ~153 GB/s each, no switch):
  * weights travel as a few large flat buckets (one broadcast per <= 1 GiB of one dtype) instead of one
    collective per tensor (~310 for the AR model, ~700 for the codec): the cost of a collective here is its
    launch and ring set-up, not its bytes;
  * audio returns over direct links: every rank sends ONE exact-size buffer to rank 0, which posts all
    receives at once (7 peers feed it concurrently, each over its own link) -- no padding to the longest
    dialogue and no W x padded staging on the root."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_indices(lengths, world):
    """Deal items to ranks by estimated work (`lengths`): longest first, each to the rank with the least work so far
    among those that still have room (every rank gets floor or ceil of n / world items, so the per-rank batches stay
    equally wide).  Deterministic.  -> list of index lists."""
    n = len(lengths)
    shards = [[] for _ in range(world)]
    if not n:
        return shards
    order = sorted(range(n), key=lambda i: (-float(lengths[i]), i))
    cap = -(-n // world)
    full_ranks = n - (cap - 1) * world                       # how many ranks hold `cap` items in the end
    load = [0.0] * world
    for i in order:
        at_cap = sum(1 for s in shards if len(s) == cap)
        open_ = [r for r in range(world)
                 if len(shards[r]) < cap - 1 or (len(shards[r]) == cap - 1 and at_cap < full_ranks)]
        best = min(open_, key=lambda r: (load[r], len(shards[r]), r))
        shards[best].append(i)
        load[best] += float(lengths[i])
    return shards


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_state_dict(sd, device, src=0, bucket_bytes=1 << 30):
    """Rank `src` holds `sd` (name -> tensor); every rank yields (name, tensor on `device`), in `sd`'s order per dtype.
    Tensors of one dtype are packed into flat buckets of up to `bucket_bytes` and each bucket is ONE broadcast; the
    yielded tensors are views into the bucket (the engines copy / re-lay what they bind, so a bucket is freed as soon
    as its views are dropped).  A tensor larger than a bucket gets a bucket of its own."""
    if _world() == 1:
        for k, v in sd.items():
            yield k, v.to(device)
        return
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    by_dtype = {}
    for name, shape, dt in meta[0]:
        by_dtype.setdefault(dt, []).append((name, shape))
    for dt, entries in by_dtype.items():
        dtype = getattr(torch, dt)
        esz = torch.empty((), dtype=dtype).element_size()
        bucket, nbytes = [], 0

        def flush(bucket):
            numels = [int(torch.Size(s).numel()) for _, s in bucket]
            total = sum(numels)
            if rank == src:
                flat = torch.empty(total, dtype=dtype, device=device)
                off = 0
                for (name, _), n in zip(bucket, numels):
                    flat[off:off + n].copy_(sd[name].reshape(-1))
                    off += n
            else:
                flat = torch.empty(total, dtype=dtype, device=device)
            dist.broadcast(flat, src=src)
            off = 0
            for (name, shape), n in zip(bucket, numels):
                yield name, flat[off:off + n].view(shape)
                off += n

        for name, shape in entries:
            b = int(torch.Size(shape).numel()) * esz
            if bucket and nbytes + b > bucket_bytes:
                yield from flush(bucket)
                bucket, nbytes = [], 0
            bucket.append((name, shape))
            nbytes += b
        if bucket:
            yield from flush(bucket)


def gather_audio(local, device, dst=0):
    """local: list of (global_index, FloatTensor(n,) or None).  Returns on `dst` the list
    [(index, tensor|None)] of the whole job (sorted by index); other ranks get None.
    One all_gather of the (index, length) table, then every rank sends its samples as ONE flat fp32 buffer of exactly
    their total length straight to `dst`, which has posted all receives at once."""
    if _world() == 1:
        return sorted([(i, None if w is None else w.to(torch.float32).reshape(-1)) for i, w in local], key=lambda x: x[0])
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    max_items = int(max(int(c.item()) for c in counts))
    meta = torch.full((max(max_items, 1), 2), -1, dtype=torch.int64, device=device)   # (index, length or -1 = failed)
    for j, (idx, wav) in enumerate(local):
        meta[j, 0] = idx
        meta[j, 1] = -1 if wav is None else wav.numel()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu() for m in metas]
    totals = [int(m[:int(c.item()), 1].clamp(min=0).sum().item()) for m, c in zip(metas, counts)]
    parts = [w.to(device=device, dtype=torch.float32).reshape(-1) for _, w in local if w is not None]
    flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float32, device=device)
    if rank != dst:
        if totals[rank]:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, dst)]):
                req.wait()
        return None
    bufs = [flat if r == dst else torch.empty(totals[r], dtype=torch.float32, device=device) for r in range(world)]
    ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != dst and totals[r]]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    out = []
    for r in range(world):
        off = 0
        for j in range(int(counts[r].item())):
            idx, n = int(metas[r][j, 0].item()), int(metas[r][j, 1].item())
            if n < 0:
                out.append((idx, None))
            else:
                out.append((idx, bufs[r][off:off + n].clone()))
                off += n
    return sorted(out, key=lambda x: x[0])


def gather_objects(obj, dst=0):
    """Small python objects (text metadata) to `dst`: list indexed by rank there, None elsewhere."""
    if _world() == 1:
        return [obj]
    got = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, got, dst=dst)
    return got


def process_batch_sharded(batch_items, run_local, lengths=None, device="cpu"):
    """Deal `batch_items` to ranks, run `run_local(items, global_indices)` -> list of
    FloatTensor|None on each rank, gather the audio on rank 0.  Returns (on rank 0) a list
    aligned with batch_items."""
    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    if lengths is None:
        lengths = [len(str(it.get("text", ""))) for it in batch_items]
    mine = shard_indices(lengths, world)[rank]
    wavs = run_local([batch_items[i] for i in mine], mine)
    got = gather_audio(list(zip(mine, wavs)), device)
    if got is None:
        return None
    res = [None] * len(batch_items)
    for idx, wav in got:
        res[idx] = wav
    return res
