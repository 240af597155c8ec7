"""The N>1 path on CPU: world_size 2 over gloo.  Each rank generates and 'aligns' only its own contiguous shard and
the scores are gathered in rank order; the result must equal the single-process run.  The alignment itself is
stood in for by the CPU oracle here (test only) -- what is under test is the sharding/gather plumbing bench.py uses."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the max-over-ranks timing reduction of bench.py
    assert float(t[0]) == float(world)
    np.save(os.path.join(out_dir, "scores_%d.npy" % rank), full.numpy())
    np.save(os.path.join(out_dir, "idx_%d.npy" % rank), idx.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(built, tmp_path):
    from aim_amd import engine, shard
    from oracle import oracle
    world, n = 2, 3000
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    ms, rs = engine.launcher_sizes("wfa", 100, 0.02)
    req, pat, txt = engine.gen_pairs(42, 0, world * n, 100, 0.02, rs)
    res, _, _ = oracle.align_batch(oracle.params("wfa", ms, rs, reduce=True), req["pattern_len"], req["text_len"], pat, txt)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("scores_%d.npy" % r)), res["score"])
        assert np.array_equal(np.load(tmp_path / ("idx_%d.npy" % r)), np.arange(world * n))


def test_shard_range_covers_everything():
    from aim_amd import shard
    for total in (0, 1, 7, 64, 1000, 4194304):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
