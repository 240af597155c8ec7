#!/usr/bin/env python3
"""Lanes per pair of wfa_group_kernel on short reads with MAX_SCORE 11 .. 50: kernel time for every feasible forced
AIM_GROUP_G next to the planner's own choice. G -> 1 is the "one pair per lane, window in LDS" design; what it costs is LDS per wavefront
(64 / G pairs' windows) and therefore residency.   python tools/group_g_sweep.py [l=100] [e=0.05] [n=1048576]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
l = int(sys.argv[1]) if len(sys.argv) > 1 else 100
e = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
ms, rs = engine.launcher_sizes("wfa", l, e)
req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
for bt in (False, True):
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=bt)
    ref = None
    for g in ("", "1", "2", "4", "8", "16", "32"):
        if g:
            os.environ["AIM_GROUP_G"] = g
        else:
            os.environ.pop("AIM_GROUP_G", None)
        try:
            with engine.DeviceSet(1) as s:
                s.configure(params, n)
                best = None
                for _ in range(3):
                    k0 = s.timers()[1]
                    s.push(0, req, pat, txt); s.launch()
                    k = s.timers()[1] - k0
                    best = k if best is None else min(best, k)
                res, _ = s.pull(0)
                plan = s.plan_describe(0)
                fb = s.fallback_pairs(0)
        except Exception as ex:
            print(json.dumps({"l": l, "e": e, "cigar": bt, "forced_G": g or "plan", "error": str(ex)[:200]}), flush=True)
            continue
        if ref is None:
            ref = res["score"].copy()
        print(json.dumps({"l": l, "e": e, "max_score": ms, "read_size": rs, "cigar": bt, "forced_G": g or "plan", "kernel_ms": best, "pairs_per_s": n / (best * 1e-3),
                          "same_scores": bool((res["score"] == ref).all()), "fallback_pairs": fb, "plan": plan}), flush=True)
