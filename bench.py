#!/usr/bin/env python3
"""Headline benchmark: aligned pairs/s (and GCUPS), WFA-adaptive score-only, l=100, e=1%, 4M synthetic pairs
per MI355X (BASELINE.json configs[1]); one process per GPU, pairs sharded statically (weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (one aim_align_device launch) over the rank's HBM-resident batch.
torch is used for device memory, the stream and torch.distributed only; the work is libaim_hip.so.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_PEAK_TLANEOPS = 256 * 64 * 2.4e9 / 1e12    # 256 CUs x 64 lanes per clock x 2.4 GHz = 39.3 T 32-bit lane-operations / s
ISSUE_PEAK_GINSTR = 1024 * 2.4e9 / 1e9          # 1 024 SIMDs x one wavefront instruction per clock = 2 458 G instructions / s

# BASELINE.json configs[1..4]. The N = 1 contract line is cfg2 (the configuration the metric is quoted on); the others run the
# same harness (same static split, same barrier / max-over-ranks timing) so that a multi-GPU run can cover "SWG ... sharded across
# 8 MI355X" and "GenASM ... 8 MI355X". Each names the roof that bounds ITS kernel: cfg2 streams (HBM), cfg4 is bound by the
# integer vector rate (lane-operations per DP cell), cfg3 and cfg5 by VALU instruction issue (tag "issue": priced per pair, not per cell).
CONFIGS = {
    "cfg2": dict(algo="wfa", l=100, e=0.01, n=1 << 22, bt=False, reduce=True, bound="hbm", pmc="wfa_lane",
                 name="WFA-adaptive score-only l=100 e=1%"),
    "cfg3": dict(algo="wfa", l=1000, e=0.05, n=1 << 18, bt=True, reduce=True, bound="issue", pmc="wfa_group",
                 name="WFA-adaptive with CIGAR l=1000 e=5%"),
    "cfg4": dict(algo="swg", l=10000, e=0.01, n=1024, bt=True, reduce=False, bound="valu", pmc="dp_strip",
                 name="SWG affine-gap with CIGAR l=10000 e=1%"),
    # cfg5 (round 5): 16 384 pairs per GPU as FOUR batches of 4 096 on four streams -- what the host CLI's --slots 4 does. One
    # synthetic pair in ~4 000 loses the diagonal and keeps one wavefront busy for ~27 ms next to a batch that takes ~8: a single straggler-free batch
    # (the round-4 line) overstates the steady rate; `single_batch` keeps that figure as a sub-record.
    "cfg5": dict(algo="genasm", l=100000, e=0.10, n=16384, batches=4, bt=True, reduce=False, bound="issue", pmc="genasm_wave",
                 name="GenASM bit-vector edit distance with CIGAR l=100000 e=10% (parity unpinned)"),
}


def pmc_instructions(tag, pairs=None):
    """Instruction counts per pair of a config's kernel from the newest committed PMC summary (profiles/rNN/<tag>_pmc_summary.json):
    (VALU wave-instructions per pair, all wave-instructions per pair, source file). The counts are a property of the code and
    the data, not of the run; the rate they are multiplied with is measured live."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", tag + "_pmc_summary.json"))):
        try:
            d = json.load(open(f))
            raw, n = d["raw"], d["pairs_per_launch"]
            valu = raw["SQ_INSTS_VALU"]["per_launch_mean"] / n
            allk = sum(raw[k]["per_launch_mean"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
                                                             "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH") if k in raw) / n
            best = (valu, allk, os.path.relpath(f, ROOT), d.get("derived", {}).get("valu_lane_use"))
        except Exception:
            continue
    return best


def pmc_traffic(kernel, pairs, io):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (FETCH_SIZE x2-corrected + WRITE_SIZE,
    separate passes; profiles/rNN/*_pmc_summary.json) when it was taken on this kernel and batch size."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if kernel in d.get("kernel", "") and d.get("pairs_per_launch") == pairs and d.get("io", "default") == io:
            best = (d["hbm_traffic_bytes_per_launch"], os.path.relpath(f, ROOT))
    return best


def _pinned(lib, capi, arr):
    """Copy a numpy array into pinned host memory (aim_host_alloc); returns (pointer, view)."""
    p = C.c_void_p()
    capi.check(lib.aim_host_alloc(C.byref(p), max(arr.nbytes, 1)))
    view = np.ctypeslib.as_array((C.c_uint8 * max(arr.nbytes, 1)).from_address(p.value))
    view[: arr.nbytes] = arr.view(np.uint8).reshape(-1)
    return p, view


def e2e_leg(lib, capi, engine, local_rank, ms, rs, req, pat, txt, batches, packed, backtrace):
    """PCIe-inclusive rate of the drop-in path (AIM's CPU-DPU + DPU Kernel + DPU-CPU, input already in
    pinned host memory, file parsing excluded): `batches` batches of the same pairs through aim_set_submit / aim_set_wait on
    two slots, so H2D(k+1) || kernel(k) || D2H(k-1). packed = 2 bits per base + raw side list; else ASCII rows."""
    n = len(req)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=backtrace, req8=True, res8=not backtrace)
    req8 = engine.to_request8(req)
    keep = []
    io = [capi.BatchIO(), capi.BatchIO()]
    runs_cap = 8 * n if backtrace else 0
    h2d_bytes = req8.nbytes
    if packed:
        pp, pt, raw, rawp, rawt = engine.pack_batch_native(params, req, pat, txt, threads=min(64, os.cpu_count() or 1))
        h2d_bytes += pp.nbytes + pt.nbytes + raw.nbytes + rawp.nbytes + rawt.nbytes
    else:
        h2d_bytes += pat.nbytes + txt.nbytes
    for k in range(2):
        b = io[k]
        b.n_pairs = n
        ptr, v = _pinned(lib, capi, req8); keep.append(v); b.requests = ptr.value
        if packed:
            for name, arr in (("packed_patterns", pp), ("packed_texts", pt), ("raw_pairs", raw), ("raw_patterns", rawp), ("raw_texts", rawt)):
                ptr, v = _pinned(lib, capi, arr); keep.append(v); setattr(b, name, ptr.value)
            b.n_raw = len(raw)
        else:
            for name, arr in (("patterns", pat), ("texts", txt)):
                ptr, v = _pinned(lib, capi, arr); keep.append(v); setattr(b, name, ptr.value)
        if backtrace:
            ptr, v = _pinned(lib, capi, np.zeros(n, dtype=capi.CIGAR_DTYPE)); keep.append(v); b.cigars = ptr.value
            ptr, v = _pinned(lib, capi, np.zeros(runs_cap, dtype=np.uint32)); keep.append(v); b.runs = ptr.value
            b.runs_cap = runs_cap
        else:
            ptr, v = _pinned(lib, capi, np.zeros(n, dtype=capi.RESULT8_DTYPE)); keep.append(v); b.results = ptr.value
    s = C.c_void_p()
    ids = (C.c_int * 1)(local_rank)
    capi.check(lib.aim_set_alloc(1, ids, C.byref(s)))
    capi.check(lib.aim_set_configure_slots(s, C.byref(params), n, 2, max(len(raw), 1) if packed else 0, runs_cap))
    nr = C.c_uint32()
    total_runs = 0
    for rep in range(2):                       # first pass warms up (first-touch of device buffers, clocks)
        t0 = time.perf_counter()
        capi.check(lib.aim_set_submit(s, 0, 0, C.byref(io[0])))
        for k in range(batches):
            if k + 1 < batches:
                capi.check(lib.aim_set_submit(s, 0, (k + 1) & 1, C.byref(io[(k + 1) & 1])))
            capi.check(lib.aim_set_wait(s, 0, k & 1, C.byref(nr)))
            total_runs = nr.value
        dt = time.perf_counter() - t0
    h2d, kern, d2h = C.c_float(), C.c_float(), C.c_float()
    lib.aim_set_timers(s, C.byref(h2d), C.byref(kern), C.byref(d2h))
    lib.aim_set_free(s)
    d2h_bytes = (16 * n + 4 * total_runs) if backtrace else 8 * n
    return {"pairs_per_s": batches * n / dt, "batches": batches, "pairs_per_batch": n, "slots": 2,
            "input": "packed 2 bit/base + raw side list" if packed else "ASCII rows",
            "output": "device-side CIGAR runs" if backtrace else "{idx, score}",
            "h2d_bytes_per_pair": h2d_bytes / n, "d2h_bytes_per_pair": d2h_bytes / n,
            "h2d_GBps_effective": batches * h2d_bytes / dt / 1e9,
            "phase_ms_per_batch": {"h2d": h2d.value / (2 * batches), "kernel": kern.value / (2 * batches), "d2h": d2h.value / (2 * batches)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2", help="BASELINE.json configuration (default and contract line: cfg2)")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU (default: the configuration's; cfg2 = 4M)")
    ap.add_argument("--batches", type=int, default=None, help="launches in flight per step, each on its own stream (default: the configuration's; cfg5 = 4)")
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--error", type=float, default=None)
    ap.add_argument("--backtrace", action="store_true", help="also produce CIGAR ops (not the headline config)")
    ap.add_argument("--io", choices=["compact", "default"], default="compact",
                    help="wire layout of requests/results: 'compact' = the reference's own 8-B WFA request_t + 8-B {idx, score} "
                         "results (AIM_FLAG_REQ8|RES8, score-only); 'default' = the 16-B / 24-B NW/SWG structs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-io", action="store_true", help="skip the extra default-layout timing (profiling runs: one kernel shape only)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive leg (reported as `e2e`, never as `value`)")
    ap.add_argument("--verify-pairs", type=int, default=1 << 20, help="pairs re-checked against the CPU oracle after timing")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (RCCL) and run the final gather even at world size 1 (also AIM_BENCH_FORCE_DIST=1): "
                         "exercises the N > 1 exchange path on a single-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the alignment path has no CPU fallback", file=sys.stderr)
        sys.exit(1)
    # Test hook only (single-GPU boxes): AIM_BENCH_SHARE_GPU=1 puts every rank on device 0 and uses gloo, so the
    # multi-rank control flow can be exercised without N GPUs.  Never set by the driver.
    share = os.environ.get("AIM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = args.force_dist or os.environ.get("AIM_BENCH_FORCE_DIST") == "1"
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:   # not under torch.distributed.run: a group of this one process
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    from aim_amd import capi, engine, shard
    lib = capi.load()

    cfg = CONFIGS[args.config]
    algo = cfg["algo"]
    args.pairs = args.pairs if args.pairs is not None else cfg["n"]
    args.length = args.length if args.length is not None else cfg["l"]
    args.error = args.error if args.error is not None else cfg["e"]
    args.backtrace = args.backtrace or cfg["bt"]
    headline = args.config == "cfg2"
    n = args.pairs
    if algo == "genasm":   # no penalties, no score cap: MAX_SCORE is ignored; READ_SIZE by the launchers' rule
        ms, rs = 0, int(np.ceil((args.length + args.length * args.error + 7) / 8)) * 8
    else:
        ms, rs = engine.launcher_sizes(algo, args.length, args.error)
    compact = args.io == "compact" and headline
    req8, res8 = compact, compact and not args.backtrace
    params = engine.make_params(algo, ms, rs, reduce=cfg["reduce"], backtrace=args.backtrace, req8=req8, res8=res8)
    res_dtype = capi.RESULT8_DTYPE if res8 else capi.RESULT_DTYPE
    # static contiguous split: rank r owns global pairs [r*n, (r+1)*n)  (host.c:191-209)
    req, pat, txt = engine.gen_pairs(42, shard.weak_first_index(n, rank), n, args.length, args.error, rs)
    alg_bytes = int(req["pattern_len"].astype(np.int64).sum() + req["text_len"].astype(np.int64).sum() + 16 * n)
    cells = int((req["pattern_len"].astype(np.int64) * req["text_len"].astype(np.int64)).sum())

    def to_dev(a, pad=64):
        t = torch.zeros(a.nbytes + pad, dtype=torch.uint8, device=dev)
        t[: a.nbytes].copy_(torch.from_numpy(a.view(np.uint8).reshape(-1)))
        return t

    d_req, d_pat, d_txt = to_dev(engine.to_request8(req) if req8 else req), to_dev(pat), to_dev(txt)
    d_res = torch.zeros(n * res_dtype.itemsize + 64, dtype=torch.uint8, device=dev)
    d_ops = torch.zeros(n * 2 * rs + 64, dtype=torch.uint8, device=dev) if args.backtrace else None
    stream = torch.cuda.current_stream(dev)
    # a step = one pass over the rank's n pairs: one launch, or (cfg5) `batches` launches of n / batches pairs in flight together, each on its own
    # stream with its own scratch, forked from and joined into the timed stream by events (the HIP events on that stream bracket all of them)
    nb = args.batches if args.batches else (int(cfg.get("batches", 1)) if args.pairs == cfg["n"] else 1)
    if nb < 1 or n % nb:
        nb = 1
    B = n // nb
    scratch_bytes = lib.aim_scratch_bytes(C.byref(params), B)
    d_scratches = [torch.zeros(max(scratch_bytes, 256), dtype=torch.uint8, device=dev) for _ in range(nb)]
    d_scratch = d_scratches[0]
    side = [torch.cuda.Stream(dev) for _ in range(nb)] if nb > 1 else []
    req_b = (8 if req8 else 16)

    def launch_batch(b, st, count=None):
        capi.check(lib.aim_align_device(C.byref(params), B if count is None else count, d_req.data_ptr() + b * B * req_b, d_pat.data_ptr() + b * B * rs,
                                        d_txt.data_ptr() + b * B * rs, d_res.data_ptr() + b * B * res_dtype.itemsize,
                                        (d_ops.data_ptr() + b * B * 2 * rs) if d_ops is not None else None,
                                        d_scratches[b].data_ptr(), d_scratches[b].numel(), st.cuda_stream))

    def step():
        if nb == 1:
            launch_batch(0, stream)
            return
        fork = torch.cuda.Event()
        fork.record(stream)
        for b in range(nb):
            side[b].wait_event(fork)
            launch_batch(b, side[b])
            done = torch.cuda.Event()
            done.record(side[b])
            stream.wait_event(done)

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream
    single_batch = None
    if nb > 1:   # the first batch alone (the round-4 figure): a sub-record, never `value`
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        launch_batch(0, stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        single_batch = {"pairs": B, "kernel_ms": e0.elapsed_time(e1), "pairs_per_s_per_gpu": B / (e0.elapsed_time(e1) * 1e-3),
                        "note": "batch 0 alone on one stream; not `value`"}
    if dist_on:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # the same kernel on the DEFAULT wire layouts (16-B requests, 24-B results: what round 1's headline used), so that a change of
    # `value` between rounds can be split into kernel and I/O-definition. Untimed by the contract; rank 0's device only.
    default_io = None
    if headline and compact and not args.backtrace and rank == 0 and not args.no_default_io:
        p2 = engine.make_params(algo, ms, rs, reduce=cfg["reduce"])
        d_req2 = to_dev(req)
        d_res2 = torch.zeros(n * capi.RESULT_DTYPE.itemsize + 64, dtype=torch.uint8, device=dev)
        def step2():
            capi.check(lib.aim_align_device(C.byref(p2), n, d_req2.data_ptr(), d_pat.data_ptr(), d_txt.data_ptr(), d_res2.data_ptr(), None,
                                            d_scratch.data_ptr(), d_scratch.numel(), stream.cuda_stream))
        for _ in range(2):
            step2()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            step2()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        k2 = e0.elapsed_time(e1) / args.steps
        default_io = {"kernel_ms": k2, "pairs_per_s_per_gpu": n / (k2 * 1e-3), "wire_bytes_per_pair": 2 * rs + 16 + 24,
                      "roofline_frac": alg_bytes / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "same kernel, default 16-B request / 24-B result structs; not `value`"}
        del d_req2, d_res2

    # final score / CIGAR gather to every rank over RCCL/xGMI (host.c:316-327) -- outside the timed region, reported separately
    gather_ms, gather = None, None
    res_host = np.frombuffer(d_res[: n * res_dtype.itemsize].cpu().numpy().tobytes(), dtype=res_dtype)
    ops_host = d_ops[: n * 2 * rs].cpu().numpy().reshape(n, 2 * rs) if args.backtrace else None
    if dist_on:
        scores = torch.from_numpy(np.ascontiguousarray(res_host["score"])).to(dev)
        torch.cuda.synchronize(dev)
        g0 = time.perf_counter()
        out = shard.gather_scores(scores, dist, force=True)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert bool((out[rank * n:(rank + 1) * n] == scores).all())
        gather = {"backend": dist.get_backend(), "world": world, "scores_ms": gather_ms, "scores_bytes": int(out.numel() * out.element_size())}
        if args.backtrace:
            # with CIGAR the exchange carries the compact form: 16-B aim_cigar_t per pair + its runs, made by the drop-in path
            # (aim_set_submit with a run buffer) from the rank's own batch; run counts differ per rank (shard.gather_cigars)
            del d_scratch
            torch.cuda.empty_cache()
            runs_cap = min(n * max(8, rs // 4 + 2), 1 << 28)
            try:
                with engine.DeviceSet(1, [local_rank]) as s:
                    s.configure_slots(params, n, slots=1, max_raw=0, max_runs=runs_cap)
                    s.submit(0, 0, req, pat, txt, cigar_runs_cap=runs_cap)
                    cg = s.wait(0, 0)
                cig_d = torch.from_numpy(cg["cig"].view(np.int32).reshape(n, 4).copy()).to(dev)
                runs_d = torch.from_numpy(cg["runs"].view(np.int32).copy()).to(dev)
                torch.cuda.synchronize(dev)
                g0 = time.perf_counter()
                cig_all, runs_all, counts = shard.gather_cigars(cig_d, runs_d, dist, force=True)
                torch.cuda.synchronize(dev)
                cigar_ms = (time.perf_counter() - g0) * 1e3
                # this rank's slice of what every rank now holds prints like its own ops rows (edit_cigar_print, host.c:69-89)
                nchk = min(n, 4096)
                mine = cig_all[rank * n: rank * n + nchk].cpu().numpy().view(np.uint32).reshape(-1).view(capi.CIGAR_DTYPE)
                same = engine.format_output_runs(mine, runs_all.cpu().numpy().view(np.uint32)) == engine.format_output(res_host[:nchk], ops_host[:nchk], True)
                gather.update({"cigar_ms": cigar_ms, "cigar_bytes": int(cig_all.numel() * 4 + runs_all.numel() * 4), "runs_per_rank": counts,
                               "cigar_matches_ops_rows": bool(same)})
                assert same
            except capi.AimError as e:   # (a run buffer too small for this configuration's CIGARs: reported, scores were still gathered)
                gather["cigar_skipped"] = str(e)

    # correctness of what was timed: re-check a bounded prefix against the CPU oracle (checker only)
    from oracle import oracle
    nv = min(n, args.verify_pairs, max(8, int(2e9 // max(1, (args.length * args.length if algo != "wfa" else args.length * 200)))))
    op = oracle.params(algo, ms, rs, reduce=cfg["reduce"], backtrace=False)
    ores, _, worst = oracle.align_batch(op, req["pattern_len"][:nv], req["text_len"][:nv], pat[:nv], txt[:nv],
                                        nthreads=os.cpu_count() or 1)
    verified = bool(worst == 0 and np.array_equal(ores["score"], res_host["score"][:nv])
                    and np.array_equal(res_host["idx"], req["idx"]))
    if dist_on:
        v = torch.tensor([1 if verified else 0], device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(int(v[0]))

    # GenASM, long reads: a pair whose windowed traceback loses the diagonal has every later window random against random (the full-width
    # path) and keeps ONE wavefront busy several times longer than the batch takes (profiles/NOTES.md R4.6). Say whether this
    # rank's batch holds one: its score is ~half its length instead of ~the error rate.
    tail = None
    if algo == "genasm":
        sc = res_host["score"].astype(np.int64)
        med = float(np.median(sc)) if len(sc) else 0.0
        lost = np.nonzero(sc > 2 * med + 64)[0]
        tail = {"pairs_that_lost_the_diagonal": int(len(lost)), "first": [int(i) for i in lost[:4]], "median_score": med,
                "batches": nb, "batches_holding_one": int(len(set(int(i) // B for i in lost))),
                "note": "synthetic l=100000 e=10%: ~1 pair in 4000; a batch with one takes ~27 ms instead of ~8 (tools/ga_tail.py); `value` counts every batch"}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        passes, spent = 0, 0.0
        nb = n if headline else min(n, max(cores, nv))      # long-read configs: a bounded sample of the batch (one pair per thread at least)
        while spent < 3.0 and passes < 64:
            c0 = time.perf_counter()
            oracle.align_batch(op, req["pattern_len"][:nb], req["text_len"][:nb], pat[:nb], txt[:nb], nthreads=cores)
            spent += time.perf_counter() - c0
            passes += 1
        cpu_baseline = {"value": passes * nb / spent, "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": "%d passes over %s %d pairs of the batch, %d threads, %.1f s wall (oracle/%s, score-only)"
                                  % (passes, "the same" if nb == n else "the first", nb, cores, spent,
                                     "genasm_oracle.c" if algo == "genasm" else "aim_oracle.c")}

    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e and headline:
        ne = min(n, 1 << 22)
        e2e = {"packed": e2e_leg(lib, capi, engine, local_rank, ms, rs, req[:ne], pat[:ne], txt[:ne], 8, True, args.backtrace),
               "ascii": e2e_leg(lib, capi, engine, local_rank, ms, rs, req[:ne], pat[:ne], txt[:ne], 4, False, args.backtrace),
               "note": "input already in pinned host memory; file parsing / packing excluded (SURVEY 8d ii); never used as `value`"}

    if args.backtrace:   # algorithmic bytes with CIGAR include the ops actually produced
        alg_bytes += int((res_host["end_offset"].astype(np.int64) - res_host["begin_offset"]).sum())
    if rank == 0:
        total_pairs = world * n * args.steps
        value = total_pairs / elapsed
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        kname = lib.aim_kernel_name(C.byref(params)).decode()
        plan_buf = C.create_string_buffer(512)
        capi.check(lib.aim_plan_describe(C.byref(params), B, plan_buf, len(plan_buf)))
        traffic = pmc_traffic(kname, n, args.io + ("+cigar" if args.backtrace else "")) if headline else None   # (the CIGAR instantiation has a summary of its own)
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                    "traffic_unit": "bytes/launch", "traffic_source": traffic[1] if traffic else None,
                    "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_pair": alg_bytes / n}
        if cfg["bound"] != "hbm":
            # not an HBM-bound kernel: price it against the roof that does bound it. Instruction counts per pair come from the
            # committed PMC summary of this kernel (a property of code + data); the pair rate is this run's.
            ins = pmc_instructions(cfg["pmc"])
            rate = n / (kernel_ms * 1e-3)                      # pairs / s of one GPU's kernel
            hbm = {k: roofline[k] for k in ("achieved", "frac", "algorithmic_bytes_per_pair")}
            if cfg["bound"] == "valu":
                ops_per_cell = ins[0] * 64.0 / (cells / n) if ins else None
                ach = rate * ins[0] * 64.0 / 1e12 if ins else None
                roofline = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                            "frac": ach / VALU_PEAK_TLANEOPS if ins else None, "lane_ops_per_cell": ops_per_cell,
                            "instruction_source": ins[2] if ins else None, "traffic": None, "hbm_view": hbm}
            else:
                # WFA / GenASM wavefronts: instruction-bound. A wave64 VALU instruction occupies its SIMD for 4 clocks, so the vector
                # unit saturates at a quarter of the all-types issue rate: the VALU fraction is the one that says how full the SIMDs
                # are (cfg5 at 4 096 pairs: 0.97) and is `frac`; the all-types issue rate VERDICT r02 asked for stays as `issue_view`.
                ach = rate * ins[0] * 64.0 / 1e12 if ins else None
                roofline = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                            "frac": ach / VALU_PEAK_TLANEOPS if ins else None, "valu_instructions_per_pair": ins[0] if ins else None,
                            # `frac` says how busy the SIMDs are issuing vector instructions; this says how many of the 64 lanes those instructions
                            # had switched on (PMC: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)). Their product is the useful fraction.
                            "useful_lane_frac": ins[3] if ins else None,
                            "frac_useful": (ach / VALU_PEAK_TLANEOPS * ins[3]) if (ins and ins[3]) else None,
                            "instruction_source": ins[2] if ins else None, "traffic": None, "hbm_view": hbm,
                            "issue_view": {"achieved": rate * ins[1] / 1e9, "peak": ISSUE_PEAK_GINSTR, "unit": "G wavefront-instructions/s",
                                           "frac": rate * ins[1] / 1e9 / ISSUE_PEAK_GINSTR, "instructions_per_pair": ins[1]} if ins else None}
        line = {
            "metric": ("aligned pairs/sec WFA-adaptive l=%d e=%g%%" % (args.length, args.error * 100)) if headline else
                      ("aligned pairs/sec %s" % cfg["name"]),
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": ("WFA-adaptive %s l=%d e=%g%% %d synthetic pairs per GPU (MAX_SCORE %d, READ_SIZE %d)"
                                    % ("with CIGAR" if args.backtrace else "score-only", args.length, args.error * 100, n, ms, rs)) if headline else
                                   ("%s: %s, %d synthetic pairs per GPU (MAX_SCORE %d, READ_SIZE %d)" % (args.config, cfg["name"], n, ms, rs)),
                       "pairs_per_gpu": n, "parallelism": "pairs sharded statically, %d rank(s)" % world,
                       "kernel": kname, "plan": plan_buf.value.decode(),
                       "io": ("compact: 8-B WFA request_t (common.h:172-177) + 8-B {idx, score} results" if res8 else
                              ("8-B requests, 24-B results" if req8 else "default: 16-B requests, 24-B results")),
                       "wire_bytes_per_pair": 2 * rs + (8 if req8 else 16) + (8 if res8 else 24) + (2 * rs if args.backtrace else 0)},
            "gcups": value * (cells / n) / 1e9,
            "kernel_ms": kernel_ms,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "default_io": default_io,
            "e2e": e2e,
            "gather_ms": gather_ms,
            "gather": gather,
            "verified_vs_oracle": verified,
            "tail": tail,
            "single_batch": single_batch,
        }
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if not verified:
        sys.exit(3)


if __name__ == "__main__":
    main()
