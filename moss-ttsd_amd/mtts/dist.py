"""Batch sharding over the GPUs of one node: dialogues are independent, so the
JSONL batch is dealt to ranks, every rank runs the full engine, and the only exchanges are a
weight broadcast at start-up and a length/audio gather at the end (RCCL over xGMI on GPUs, gloo
in the CPU tests).  No per-step collective, no tensor parallelism."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_indices(lengths, world):
    """Longest-first round-robin deal: balances KV growth per rank.  -> list of index lists."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world)]
    for k, i in enumerate(order):
        r = k % world if (k // world) % 2 == 0 else world - 1 - (k % world)     # snake order
        shards[r].append(i)
    return shards


def broadcast_state_dict(sd, device, src=0):
    """Rank `src` holds `sd` (name -> tensor); every rank yields (name, tensor on `device`).
    One broadcast per tensor, largest first would not matter: xGMI is point-to-point and the
    root feeds 7 peers concurrently."""
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    for name, shape, dt in meta[0]:
        dtype = getattr(torch, dt)
        t = sd[name].to(device=device, dtype=dtype).contiguous() if rank == src else torch.empty(shape, dtype=dtype, device=device)
        dist.broadcast(t, src=src)
        yield name, t


def gather_audio(local, device, dst=0):
    """local: list of (global_index, FloatTensor(n,) or None).  Returns on `dst` the list
    [(index, tensor|None)] of the whole job (sorted by index); other ranks get None.
    all_gather of int64 lengths, then one padded gather of fp32 samples."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    max_items = int(max(int(c.item()) for c in counts))
    meta = torch.full((max_items, 2), -1, dtype=torch.int64, device=device)        # (index, length or -1 = failed)
    for j, (idx, wav) in enumerate(local):
        meta[j, 0] = idx
        meta[j, 1] = -1 if wav is None else wav.numel()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_len = int(max(int(m[:, 1].max().item()) for m in metas)) if max_items else 0
    max_len = max(max_len, 1)
    buf = torch.zeros(max_items, max_len, dtype=torch.float32, device=device)
    for j, (_, wav) in enumerate(local):
        if wav is not None:
            buf[j, :wav.numel()] = wav.to(device=device, dtype=torch.float32).reshape(-1)
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, bufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        for j in range(int(counts[r].item())):
            idx, n = int(metas[r][j, 0].item()), int(metas[r][j, 1].item())
            out.append((idx, None if n < 0 else bufs[r][j, :n].clone()))
    return sorted(out, key=lambda x: x[0])


def process_batch_sharded(batch_items, run_local, lengths=None, device="cpu"):
    """Deal `batch_items` to ranks, run `run_local(items, global_indices)` -> list of
    FloatTensor|None on each rank, gather the audio on rank 0.  Returns (on rank 0) a list
    aligned with batch_items."""
    world, rank = dist.get_world_size(), dist.get_rank()
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
