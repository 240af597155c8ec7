"""bench.py's contract on a GPU box: the one-JSON-line output at N = 1, and the N = 2 control flow (rendezvous, barrier,
max-over-ranks timing, score gather) under torch.distributed.run with both ranks on the one GPU a test box has
(AIM_BENCH_SHARE_GPU=1: gloo instead of RCCL -- two ranks cannot share a device under RCCL; never set by the driver)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline")


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_line(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "262144", "--steps", "3", "--warmup", "1",
                        "--verify-pairs", "65536"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["dtype"] == "int16"
    assert d["vs_baseline"] is None and d["value"] > 0 and d["verified_vs_oracle"] is True
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["unit"] == "GB/s"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and "plan" in d["config"]
    assert d["e2e"]["packed"]["pairs_per_s"] > 0            # PCIe-inclusive leg: reported, never `value`


def test_bench_two_ranks_control_flow(built):
    env = dict(os.environ, AIM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "131072", "--steps", "3",
                        "--warmup", "1", "--verify-pairs", "32768"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["verified_vs_oracle"] is True
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1      # rank 0 prints the one line
    assert d["config"]["pairs_per_gpu"] == 131072 and d["cpu_baseline"] is None and d["e2e"] is None   # rank 0 at N = 1 only
    assert d["gather_ms"] is not None
    # a mismatch between --gpus and WORLD_SIZE is refused
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 2


def test_bench_other_baseline_configs_run_through_the_same_harness(built):
    """`bench.py --config cfg3 | cfg4 | cfg5`: the other BASELINE configurations through the same static split / timing / one-line
    contract, each priced against the roof that bounds its kernel; cfg3 also under two ranks."""
    for cfg, pairs, bound in (("cfg3", "4096", "valu"), ("cfg4", "16", "valu"), ("cfg5", "64", "valu"), ("cfg5", "256", "valu")):
        extra = ["--batches", "4"] if pairs == "256" else []                 # cfg5's own form: four batches in flight on four streams
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--pairs", pairs, "--steps", "2", "--warmup", "1",
                            "--length", {"cfg3": "1000", "cfg4": "2000", "cfg5": "5000"}[cfg]] + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, (cfg, r.stderr[-2000:])
        d = _last_json(r.stdout)
        assert d["verified_vs_oracle"] is True and d["value"] > 0 and d["roofline"]["bound"] == bound and cfg in d["config"]["workload"]
        assert d["cpu_baseline"]["value"] > 0
        if cfg == "cfg5":    # round 4: lanes switched on next to SIMD time, and whether the batch holds a pair that lost the diagonal
            assert d["tail"]["pairs_that_lost_the_diagonal"] == 0 and d["tail"]["median_score"] > 0
            assert d["tail"]["batches"] == (4 if extra else 1) and (d["single_batch"] is not None) == bool(extra)
            assert d["roofline"]["useful_lane_frac"] is None or 0 < d["roofline"]["useful_lane_frac"] <= 1
        else:
            assert d["tail"] is None
    env = dict(os.environ, AIM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29732", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg3", "--pairs", "2048", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["verified_vs_oracle"] is True and d["roofline"]["bound"] == "valu" and d["roofline"]["issue_view"]["frac"] > 0 and d["gather_ms"] is not None


def test_bench_rccl_branch_at_world_size_one(built):
    """the `nccl` (= RCCL) branch of bench.py -- init_process_group(backend="nccl", device_id=...), barrier, the
    max-over-ranks all_reduce, all_gather_into_tensor of the scores and, with --backtrace, of the compact CIGAR (run counts, headers,
    padded run buffers: shard.gather_cigars) -- executed on the box's one GPU in a group of one rank under torch.distributed.run."""
    env = dict(os.environ, AIM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("AIM_BENCH_SHARE_GPU", None)
    for extra, port in ((["--backtrace"], "29741"), ([], "29742")):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pairs", "262144", "--steps", "3",
                            "--warmup", "1", "--verify-pairs", "32768", "--no-cpu-baseline", "--no-e2e"] + extra,
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        d = _last_json(r.stdout)
        assert d["n_gpus"] == 1 and d["verified_vs_oracle"] is True and d["gather_ms"] is not None and d["gather_ms"] > 0
        assert d["gather"]["backend"] == "nccl" and d["gather"]["scores_bytes"] == 4 * 262144
        if extra:
            assert d["gather"]["cigar_matches_ops_rows"] is True and d["gather"]["cigar_ms"] > 0
            assert d["gather"]["cigar_bytes"] >= 16 * 262144 + 4 * d["gather"]["runs_per_rank"][0] > 16 * 262144
    # cfg3 (wfa_group + traceback kernel, compact CIGAR from the RLE kernel) through the same exchange, two ranks sharing the GPU over gloo
    env = dict(os.environ, AIM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29743", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "cfg3", "--pairs", "2048", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _last_json(r.stdout)
    assert d["gather"]["backend"] == "gloo" and d["gather"]["cigar_matches_ops_rows"] is True and len(d["gather"]["runs_per_rank"]) == 2
