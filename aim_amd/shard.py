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


def gather_scores(local_scores, dist=None, force=False):
    """All-gather equally sized per-rank score tensors into one tensor in rank (= input) order. `force`: run the collective
    even in a group of one rank (bench.py's AIM_BENCH_FORCE_DIST: the RCCL branch on a single-GPU box)."""
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
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


def gather_cigars(cig, runs, dist=None, force=False):
    """The CIGAR half of the path's final gather (the reference gathers results + ops rows, host.c:316-327): every rank holds its pairs'
    compact CIGAR -- `cig` int32 [n, 4] = aim_cigar_t {idx, score, run_offset, n_runs | status << 16} and `runs` int32 [r_rank], r_rank
    differing per rank -- and receives all ranks' in rank (= input) order: run counts first (one all-gather of a word), then the
    headers (equal sizes) and the run buffers padded to the longest one (all_gather_into_tensor wants equal sizes; the padding is
    dropped again), with every rank's run_offset rebased onto the concatenated run buffer.
    Returns (cig_all [world * n, 4], runs_all [sum r], runs_per_rank list)."""
    import torch
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return cig.clone(), runs.clone(), [int(runs.numel())]
    world = dist.get_world_size()
    gloo = dist.get_backend() == "gloo"

    def all_gather(t):
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if gloo:
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            torch.stack(parts, out=out)
        else:
            dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1))
        return out

    # run counts AND header counts first: the header gather below wants equal sizes on every rank (a shorter last shard would hang the collective or
    # shift offsets silently -- ADVICE r04), and the rebased run offsets are int32
    both = all_gather(torch.tensor([runs.numel(), cig.shape[0]], dtype=torch.int64, device=runs.device)).view(world, 2).tolist()
    counts = [int(c[0]) for c in both]
    if any(int(c[1]) != int(cig.shape[0]) for c in both):
        raise ValueError("gather_cigars: every rank must pass the same number of CIGAR headers (got %s); pad the last shard" % [int(c[1]) for c in both])
    if sum(counts) >= 2 ** 31:
        raise OverflowError("gather_cigars: %d runs over all ranks do not fit the int32 run_offset" % sum(counts))
    longest = max(max(counts), 1)
    padded = torch.zeros(longest, dtype=runs.dtype, device=runs.device)
    padded[: runs.numel()] = runs
    runs_by_rank = all_gather(padded)
    cig_by_rank = all_gather(cig)
    base = 0
    for r in range(world):
        cig_by_rank[r, :, 2] += base          # run_offset (values stay below 2^31: a run buffer is at most 2^28 runs per rank)
        base += counts[r]
    runs_all = torch.cat([runs_by_rank[r, : counts[r]] for r in range(world)])
    return cig_by_rank.view(-1, cig.shape[1]), runs_all, counts
