"""The N>1 path on CPU: world_size 2 over gloo.  Each rank generates and 'aligns' only its own contiguous shard and
the scores are gathered in rank order; the result must equal the single-process run.  The alignment itself is
stood in for by the CPU oracle here (test only) -- what is under test is the sharding/gather plumbing bench.py uses."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _compact(req, res, ops):
    """aim_cigar_t headers + run buffer ((length << 8) | op) from result_t + ops rows: edit_cigar_print's loop (host.c:69-89)."""
    from aim_amd import capi
    cig = np.zeros(len(res), dtype=capi.CIGAR_DTYPE)
    runs = []
    for i in range(len(res)):
        row = ops[i, int(res["begin_offset"][i]): int(res["end_offset"][i])]
        cut = np.flatnonzero(np.diff(row)) + 1
        starts, ends = np.concatenate(([0], cut)), np.concatenate((cut, [len(row)]))
        cig[i] = (req["idx"][i], res["score"][i], len(runs), len(starts), 0)   # (global pair index, assigned at parse time)
        runs += [(int(e - b) << 8) | int(row[b]) for b, e in zip(starts, ends)]
    return cig, np.array(runs, dtype=np.uint32)


def _worker(rank, world, port, n_per_rank, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from aim_amd import engine, shard
    from oracle import oracle
    ms, rs = engine.launcher_sizes("wfa", 100, 0.02)
    req, pat, txt = engine.gen_pairs(42, shard.weak_first_index(n_per_rank, rank), n_per_rank, 100, 0.02, rs)
    res, _, worst = oracle.align_batch(oracle.params("wfa", ms, rs, reduce=True), req["pattern_len"], req["text_len"], pat, txt)
    assert worst == 0
    local = torch.from_numpy(np.ascontiguousarray(res["score"]))
    full = shard.gather_scores(local, dist)
    idx = shard.gather_scores(torch.from_numpy(req["idx"].astype(np.int64)), dist)
    # the CIGAR half of the exchange: compact CIGAR (aim_cigar_t + runs, run counts differ per rank) of this rank's pairs
    bres, bops, _ = oracle.align_batch(oracle.params("wfa", ms, rs, reduce=True, backtrace=True), req["pattern_len"], req["text_len"], pat, txt)
    cig, runs = _compact(req, bres, bops)
    cig_all, runs_all, counts = shard.gather_cigars(torch.from_numpy(cig.view(np.int32).reshape(-1, 4).copy()), torch.from_numpy(runs.view(np.int32).copy()), dist)
    assert counts[rank] == len(runs) and len(counts) == world
    np.save(os.path.join(out_dir, "cig_%d.npy" % rank), cig_all.numpy())
    np.save(os.path.join(out_dir, "runs_%d.npy" % rank), runs_all.numpy())
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the max-over-ranks timing reduction of bench.py
    assert float(t[0]) == float(world)
    np.save(os.path.join(out_dir, "scores_%d.npy" % rank), full.numpy())
    np.save(os.path.join(out_dir, "idx_%d.npy" % rank), idx.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(built, tmp_path):
    from aim_amd import capi, engine, shard
    from oracle import oracle
    world, n = 2, 3000
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    ms, rs = engine.launcher_sizes("wfa", 100, 0.02)
    req, pat, txt = engine.gen_pairs(42, 0, world * n, 100, 0.02, rs)
    res, _, _ = oracle.align_batch(oracle.params("wfa", ms, rs, reduce=True), req["pattern_len"], req["text_len"], pat, txt)
    bres, bops, _ = oracle.align_batch(oracle.params("wfa", ms, rs, reduce=True, backtrace=True), req["pattern_len"], req["text_len"], pat, txt)
    want = oracle.format_output(bres, bops, True)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("scores_%d.npy" % r)), res["score"])
        assert np.array_equal(np.load(tmp_path / ("idx_%d.npy" % r)), np.arange(world * n))
        # every rank holds the whole batch's compact CIGAR, run offsets rebased: it prints like the single-process run
        cig = np.load(tmp_path / ("cig_%d.npy" % r)).view(np.uint32).reshape(-1).view(capi.CIGAR_DTYPE)
        runs = np.load(tmp_path / ("runs_%d.npy" % r)).view(np.uint32)
        assert len(cig) == world * n and engine.format_output_runs(cig, runs) == want


def test_shard_range_covers_everything():
    from aim_amd import shard
    for total in (0, 1, 7, 64, 1000, 4194304):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
