#!/usr/bin/env python3
"""Diagnostic: where the CIGAR's time goes in nw_reg_kernel / swg_reg_kernel / dp_group_kernel -- kernel time score-only, with the direction bits made
but not stored (AIM_DEBUG_FLAGS=5), stored but not walked (=1), and complete.  Needs a diagnostic build (results are wrong with the flags set):
  python -m aim_amd.build --variant diag --flags "-DAIM_DIAG_BUILD=1";  AIM_LIB=build_ab/lib_diag.so python tools/reg_cigar_split.py [algo l e n]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aim_amd import engine

def kernel_ms(params, req, pat, txt, n, reps=3):
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(reps):
            k0 = s.timers()[1]
            s.push(0, req, pat, txt)
            s.launch()
            k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
        return best, s.plan_describe(0).split()[0]

rows = [("nw", 100, 0.01, 1 << 20, {}), ("swg", 100, 0.01, 1 << 20, {}), ("nw", 100, 0.05, 1 << 20, {}), ("swg", 100, 0.05, 1 << 20, {}),
        ("nw", 300, 0.02, 99328, {}), ("swg", 300, 0.02, 99328, dict(swg_w16=True))]
if len(sys.argv) > 4:
    rows = [(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), {})]
for algo, l, e, n, kw in rows:
    ms, rs = engine.launcher_sizes(algo, l, e)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    cells = float((req["pattern_len"].astype(np.int64) * req["text_len"]).sum())
    out = []
    for label, bt, flags in (("score-only", False, "0"), ("bits, no stores, no walk", True, "5"), ("bits + stores, no walk", True, "1"), ("complete", True, "0")):
        os.environ["AIM_DEBUG_FLAGS"] = flags
        t, kern = kernel_ms(engine.make_params(algo, ms, rs, backtrace=bt, **kw), req, pat, txt, n)
        out.append("%s %.3f ms (%.0f GCUPS)" % (label, t, cells / t / 1e6))
    os.environ["AIM_DEBUG_FLAGS"] = "0"
    print("%s l=%d e=%g n=%d %s: " % (algo, l, e, n, kern) + "; ".join(out), flush=True)
