#!/usr/bin/env python3
"""Kernel-only timings of the non-headline configurations through the C-ABI set calls (aim_set_timers' kernel
timer = AIM's "DPU Kernel").  One JSON line per configuration.  python tools/bench_configs.py [names...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine  # noqa: E402

CONFIGS = {
    "wfa_l100_e1_cigar": dict(algo="wfa", l=100, e=0.01, n=1 << 20, kw=dict(backtrace=True, reduce=True)),
    "wfa_l100_e1_x4g6a2_score": dict(algo="wfa", l=100, e=0.01, n=1 << 22, kw=dict(reduce=True), cost=dict(mismatch=4, gap_o=6, gap_e=2)),
    "wfa_l100_e1_x4g6a2_cigar": dict(algo="wfa", l=100, e=0.01, n=1 << 20, kw=dict(reduce=True, backtrace=True), cost=dict(mismatch=4, gap_o=6, gap_e=2)),
    "wfa_l100_e1_score": dict(algo="wfa", l=100, e=0.01, n=1 << 22, kw=dict(reduce=True)),
    "wfa_l100_e2_score": dict(algo="wfa", l=100, e=0.02, n=1 << 20, kw=dict(reduce=True)),
    "wfa_l100_e2_cigar": dict(algo="wfa", l=100, e=0.02, n=1 << 20, kw=dict(backtrace=True, reduce=True)),
    "wfa_l100_e5_score": dict(algo="wfa", l=100, e=0.05, n=1 << 20, kw=dict(reduce=True)),
    "wfa_l100_e10_score": dict(algo="wfa", l=100, e=0.10, n=1 << 19, kw=dict(reduce=True)),
    "wfa_l250_e5_score": dict(algo="wfa", l=250, e=0.05, n=1 << 18, kw=dict(reduce=True)),
    "wfa_l150_e2_score": dict(algo="wfa", l=150, e=0.02, n=1 << 20, kw=dict(reduce=True)),
    "wfa_l150_e1_score": dict(algo="wfa", l=150, e=0.01, n=1 << 20, kw=dict(reduce=True)),
    "wfa_l150_e1_cigar": dict(algo="wfa", l=150, e=0.01, n=1 << 20, kw=dict(backtrace=True, reduce=True)),
    "wfa_l100_e5_score_nored": dict(algo="wfa", l=100, e=0.05, n=1 << 20, kw=dict()),
    "wfa_l100_e5_cigar": dict(algo="wfa", l=100, e=0.05, n=1 << 19, kw=dict(backtrace=True, reduce=True)),
    "wfa_l1000_e5_cigar": dict(algo="wfa", l=1000, e=0.05, n=1 << 16, kw=dict(backtrace=True, reduce=True)),
    "wfa_l1000_e5_score": dict(algo="wfa", l=1000, e=0.05, n=1 << 16, kw=dict(reduce=True)),
    # WFA-adaptive on long reads: beyond wfa_group's READ_SIZE <= 2048 / MAX_SCORE <= 400 -> wfa_wave_kernel, one pair per wavefront
    "wfa_l10000_e1_score": dict(algo="wfa", l=10000, e=0.01, n=8192, kw=dict(reduce=True)),
    "wfa_l10000_e1_cigar": dict(algo="wfa", l=10000, e=0.01, n=8192, kw=dict(backtrace=True, reduce=True)),
    "wfa_l2000_e5_score": dict(algo="wfa", l=2000, e=0.05, n=1 << 14, kw=dict(reduce=True)),
    "wfa_l4000_e2_score": dict(algo="wfa", l=4000, e=0.02, n=1 << 14, kw=dict(reduce=True)),
    "nw_l100_e1_cigar": dict(algo="nw", l=100, e=0.01, n=1 << 20, kw=dict(backtrace=True)),
    "nw_l100_e1_score": dict(algo="nw", l=100, e=0.01, n=1 << 20, kw=dict()),
    "nw_l100_e5_score": dict(algo="nw", l=100, e=0.05, n=1 << 20, kw=dict()),
    "nw_l100_e5_cigar": dict(algo="nw", l=100, e=0.05, n=1 << 20, kw=dict(backtrace=True)),
    "nw_l100_e10_score": dict(algo="nw", l=100, e=0.10, n=1 << 20, kw=dict()),
    "nw_l100_e10_cigar": dict(algo="nw", l=100, e=0.10, n=1 << 20, kw=dict(backtrace=True)),
    "nw_l70_e2_score": dict(algo="nw", l=70, e=0.02, n=1 << 20, kw=dict()),
    "swg_l100_e1_cigar": dict(algo="swg", l=100, e=0.01, n=1 << 20, kw=dict(backtrace=True)),
    "swg_l100_e1_score": dict(algo="swg", l=100, e=0.01, n=1 << 20, kw=dict()),
    "swg_l100_e2_score": dict(algo="swg", l=100, e=0.02, n=1 << 20, kw=dict()),
    "swg_l100_e5_score": dict(algo="swg", l=100, e=0.05, n=1 << 20, kw=dict()),
    "swg_l100_e5_cigar": dict(algo="swg", l=100, e=0.05, n=1 << 20, kw=dict(backtrace=True)),
    "swg_l100_e10_score": dict(algo="swg", l=100, e=0.10, n=1 << 20, kw=dict()),
    "swg_l70_e2_score": dict(algo="swg", l=70, e=0.02, n=1 << 20, kw=dict()),
    # l = 150 with int16 cells (--mram / AIM_FLAG_SWG_W16): with the launcher's int8 cells EVERY pair of this length wraps (quirk S3: o + v e passes 127)
    "swg_l150_e1_w16_score": dict(algo="swg", l=150, e=0.01, n=1 << 20, kw=dict(swg_w16=True)),
    "swg_l150_e1_w16_cigar": dict(algo="swg", l=150, e=0.01, n=1 << 20, kw=dict(backtrace=True, swg_w16=True)),
    "swg_l150_e5_w16_score": dict(algo="swg", l=150, e=0.05, n=1 << 20, kw=dict(swg_w16=True)),
    "nw_l150_e1_score": dict(algo="nw", l=150, e=0.01, n=1 << 20, kw=dict()),
    "nw_l150_e1_cigar": dict(algo="nw", l=150, e=0.01, n=1 << 20, kw=dict(backtrace=True)),
    "nw_l150_e5_score": dict(algo="nw", l=150, e=0.05, n=1 << 20, kw=dict()),
    "nw_l250_e2_score": dict(algo="nw", l=250, e=0.02, n=399360, kw=dict()),
    "nw_l250_e2_cigar": dict(algo="nw", l=250, e=0.02, n=99328, kw=dict(backtrace=True)),
    "swg_l250_e2_w16_score": dict(algo="swg", l=250, e=0.02, n=399360, kw=dict(swg_w16=True)),
    "swg_l250_e2_w16_cigar": dict(algo="swg", l=250, e=0.02, n=99328, kw=dict(backtrace=True, swg_w16=True)),
    "nw_l500_e2_score": dict(algo="nw", l=500, e=0.02, n=99328, kw=dict()),
    "nw_l1200_e2_score": dict(algo="nw", l=1200, e=0.02, n=16384, kw=dict()),
    "swg_l2000_e2_w16_cigar": dict(algo="swg", l=2000, e=0.02, n=1024, kw=dict(backtrace=True, swg_w16=True)),
    "nw_l2000_e2_cigar": dict(algo="nw", l=2000, e=0.02, n=1024, kw=dict(backtrace=True)),
    # round 6: beyond READ_SIZE 2048 (two-wavefront strips of 20 / 24 / 32 cells per lane), and NW long reads below the int16 bound (READ_SIZE < 7 990: dp_strip, not the literal path)
    "nw_l2800_e5_cigar": dict(algo="nw", l=2800, e=0.05, n=1024, kw=dict(backtrace=True)),
    "swg_l2800_e5_w16_cigar": dict(algo="swg", l=2800, e=0.05, n=1024, kw=dict(backtrace=True, swg_w16=True)),
    "nw_l5000_e5_score": dict(algo="nw", l=5000, e=0.05, n=1024, kw=dict()),
    "nw_l5000_e5_cigar": dict(algo="nw", l=5000, e=0.05, n=1024, kw=dict(backtrace=True)),
    # round 6: SWG with the launchers' int8 cells (MAX_SCORE < 127: they wrap by design) beyond READ_SIZE 320: swg_lane_kernel while 64 lanes' rows fit LDS (dp_wave's one-lane literal path before)
    "swg_l500_e1_int8_score": dict(algo="swg", l=500, e=0.01, n=16384, kw=dict()),
    "swg_l1000_e1_int8_score": dict(algo="swg", l=1000, e=0.01, n=4096, kw=dict()),
    "swg_l500_e5_w16_cigar": dict(algo="swg", l=500, e=0.05, n=24576, kw=dict(backtrace=True, swg_w16=True)),   # (MAX_SCORE 125: int8 cells wrap by design here, S3 -- a pair then stops with AIM_PAIR_SWG_NO_OP)
    "swg_l1000_e5_cigar": dict(algo="swg", l=1000, e=0.05, n=1 << 12, kw=dict(backtrace=True)),
    "swg_l10000_e1_cigar": dict(algo="swg", l=10000, e=0.01, n=128, kw=dict(backtrace=True)),
    "swg_l10000_e1_cigar_n2048": dict(algo="swg", l=10000, e=0.01, n=2048, kw=dict(backtrace=True)),
    "swg_l10000_e1_cigar_n256": dict(algo="swg", l=10000, e=0.01, n=256, kw=dict(backtrace=True)),
    "nw_l1000_e5_cigar": dict(algo="nw", l=1000, e=0.05, n=1 << 12, kw=dict(backtrace=True)),
    # every text truncated to a third of its pattern (plen > 2 tlen: the last row's tail wraps around the flat table more than once; until round 6 ONE lane filled such a pair)
    "nw_l1000_literal": dict(algo="nw", l=1000, e=0.05, n=1 << 12, kw=dict(backtrace=True), text_div=3),
    "swg_l1000_literal": dict(algo="swg", l=1000, e=0.05, n=1 << 12, kw=dict(backtrace=True), text_div=3),
    "swg_l1000_e5_score": dict(algo="swg", l=1000, e=0.05, n=1 << 12, kw=dict()),
    "nw_l1000_e5_score": dict(algo="nw", l=1000, e=0.05, n=1 << 12, kw=dict()),
    "swg_l10000_e1_score_n256": dict(algo="swg", l=10000, e=0.01, n=256, kw=dict()),
    # BASELINE config 5 (parity unpinned: published GenASM algorithm, oracle/genasm_oracle.c)
    "genasm_l100000_e10_cigar": dict(algo="genasm", l=100000, e=0.10, n=1024, kw=dict(backtrace=True)),
    "genasm_l100000_e10_score": dict(algo="genasm", l=100000, e=0.10, n=1024, kw=dict()),
    "genasm_l100000_e10_cigar_n4096": dict(algo="genasm", l=100000, e=0.10, n=4096, kw=dict(backtrace=True)),
    "genasm_l100000_e10_cigar_n6144": dict(algo="genasm", l=100000, e=0.10, n=6144, kw=dict(backtrace=True)),
    "genasm_l100000_e10_cigar_n4864": dict(algo="genasm", l=100000, e=0.10, n=4864, kw=dict(backtrace=True)),
    "genasm_l100000_e10_cigar_n5120": dict(algo="genasm", l=100000, e=0.10, n=5120, kw=dict(backtrace=True)),
    "genasm_l100000_e10_score_n8192": dict(algo="genasm", l=100000, e=0.10, n=8192, kw=dict()),
    "genasm_l100000_e10_cigar_n8192": dict(algo="genasm", l=100000, e=0.10, n=8192, kw=dict(backtrace=True)),
    "genasm_l10000_e10_cigar": dict(algo="genasm", l=10000, e=0.10, n=8192, kw=dict(backtrace=True)),
    "genasm_l10000_e20_cigar": dict(algo="genasm", l=10000, e=0.20, n=8192, kw=dict(backtrace=True)),
    "genasm_l10000_e30_cigar": dict(algo="genasm", l=10000, e=0.30, n=8192, kw=dict(backtrace=True)),
    "genasm_l100_e10_cigar": dict(algo="genasm", l=100, e=0.10, n=1 << 18, kw=dict(backtrace=True)),
}


def run(name, cfg, reps=3):
    cost = cfg.get("cost", {})
    ms, rs = engine.launcher_sizes(cfg["algo"], cfg["l"], cfg["e"], **cost)
    params = engine.make_params(cfg["algo"], ms, rs, **cfg["kw"], **cost)
    req, pat, txt = engine.gen_pairs(42, 0, cfg["n"], cfg["l"], cfg["e"], rs)
    if cfg.get("text_div"):
        req["text_len"] = np.maximum(1, req["pattern_len"] // cfg["text_div"])
    with engine.DeviceSet(1) as s:
        s.configure(params, cfg["n"])
        best = None
        for _ in range(reps):
            k0 = s.timers()[1]
            s.push(0, req, pat, txt)
            s.launch()
            k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
        res, ops = s.pull(0)
        plan = s.plan_describe(0)
        todo = s.fallback_pairs(0)
    cells = float((req["pattern_len"].astype(np.int64) * req["text_len"]).sum())
    alg = float(req["pattern_len"].sum() + req["text_len"].sum() + 16 * cfg["n"])
    if ops is not None:
        alg += float((res["end_offset"] - res["begin_offset"]).sum())
    print(json.dumps({"config": name, "kernel": capi.load().aim_kernel_name(__import__("ctypes").byref(params)).decode(),
                      "pairs": cfg["n"], "max_score": ms, "read_size": rs, "kernel_ms": best,
                      "pairs_per_s": cfg["n"] / (best * 1e-3), "gcups": cells / (best * 1e-3) / 1e9,
                      "algorithmic_GBps": alg / (best * 1e-3) / 1e9, "algorithmic_bytes": alg, "mean_score": float(res["score"].mean()),
                      "todo_pairs": todo, "plan": plan}), flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or list(CONFIGS)
    for nm in names:
        try:
            run(nm, CONFIGS[nm])
        except Exception as e:   # one failing configuration must not truncate the file
            print(json.dumps({"config": nm, "error": repr(e)}), flush=True)
