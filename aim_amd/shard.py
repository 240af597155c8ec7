"""Static sharding of a pair batch over ranks and the final score gather (the only exchange step of the path).

Mirrors the reference's partition (WFA/DPU-WRAM/host/host.c:191-209): contiguous blocks in input order, no
inter-device traffic while aligning; results are concatenated in rank order.  torch.distributed is plumbing only
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import math


def shard_range(total, world, rank):
    """Pairs [lo, hi) owned by `rank` when `total` pairs are split in contiguous blocks (strong-scaling split)."""
    per = math.ceil(total / world) if world > 0 else total
    return min(total, rank * per), min(total, (rank + 1) * per)


def weak_first_index(pairs_per_rank, rank):
    """First global pair index of a rank when every rank owns `pairs_per_rank` pairs (weak scaling, bench.py)."""
    return rank * pairs_per_rank


def gather_scores(local_scores, dist=None):
    """All-gather equally sized per-rank score tensors into one tensor in rank (= input) order."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_scores.clone()
    world = dist.get_world_size()
    out = torch.empty(world * local_scores.numel(), dtype=local_scores.dtype, device=local_scores.device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(local_scores) for _ in range(world)]
        dist.all_gather(parts, local_scores)
        torch.cat(parts, out=out)
    else:
        dist.all_gather_into_tensor(out, local_scores)
    return out
