"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The path is embarrassingly image-parallel (no batch statistics anywhere: GroupNorm / LayerNorm / InstanceNorm are
per-sample), so there is NO collective inside the denoise loop.  The only collectives are
  * broadcast_weights : rank 0's parameters -> every rank, in flat ~256 MB buckets (one-shot, start-up);
  * all_gather_images : per-rank restored shards -> the full batch on every rank (one call per batch).
The reference gets the same behaviour implicitly from Lightning DDP + DistributedSampler (configs/val.yaml:11,26).
"""
from typing import List

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(model: torch.nn.Module, src: int = 0, bucket_bytes: int = 256 << 20) -> int:
    """Broadcast every parameter and buffer from `src`; returns the number of bytes moved."""
    tensors: List[torch.Tensor] = [p.data for p in model.parameters()] + [b for b in model.buffers()]
    moved = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_dtype.items():
        bucket, size = [], 0
        esize = torch.empty((), dtype=dtype).element_size()

        def flush():
            nonlocal bucket, size, moved
            if not bucket:
                return
            flat = torch.cat([t.reshape(-1) for t in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for t in bucket:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            moved += flat.numel() * esize
            bucket, size = [], 0

        for t in ts:
            bucket.append(t)
            size += t.numel() * esize
            if size >= bucket_bytes:
                flush()
        flush()
    return moved


def broadcast_weights_sharded(model: torch.nn.Module, src: int = 0, bucket_bytes: int = 512 << 20) -> int:
    """Weight broadcast as SCATTER + ALL-GATHER: xGMI is a full mesh of point-to-point links, so a
    rooted broadcast is bound by the root's links; here the root sends each rank only 1/world of a bucket (scatter) and every
    rank then forwards its slice to all others (all-gather) - every link carries 1/world of the bytes, all links busy.
    Same result as `broadcast_weights`; returns the number of payload bytes."""
    world, rank = dist.get_world_size(), dist.get_rank()
    tensors: List[torch.Tensor] = [p.data for p in model.parameters()] + [b for b in model.buffers()]
    moved = 0
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_dtype.items():
        esize = torch.empty((), dtype=dtype).element_size()
        bucket, size = [], 0

        def flush():
            nonlocal bucket, size, moved
            if not bucket:
                return
            n = sum(t.numel() for t in bucket)
            per = (n + world - 1) // world
            flat = torch.zeros(per * world, dtype=dtype, device=device)
            if rank == src:
                torch.cat([t.reshape(-1) for t in bucket], out=flat[:n])
            mine = torch.empty(per, dtype=dtype, device=device)
            dist.scatter(mine, [flat[r * per:(r + 1) * per] for r in range(world)] if rank == src else None, src=src)
            dist.all_gather_into_tensor(flat, mine)
            off = 0
            for t in bucket:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            moved += n * esize
            bucket, size = [], 0

        for t in ts:
            bucket.append(t)
            size += t.numel() * esize
            if size >= bucket_bytes:
                flush()
        flush()
    return moved


def all_gather_images(shard: torch.Tensor, sizes: List[int] = None) -> torch.Tensor:
    """Gather per-rank [b_r, C, H, W] shards into [sum b_r, C, H, W] on every rank (rank order = batch order)."""
    world = dist.get_world_size()
    shard = shard.contiguous()
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((world * shard.shape[0], *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(out, shard)
        return out
    mx = max(sizes)                                  # ragged shards: pad to the largest, gather, then trim
    padded = torch.zeros((mx, *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    padded[:shard.shape[0]] = shard
    out = torch.empty((world * mx, *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * mx:r * mx + s] for r, s in enumerate(sizes)], 0)
