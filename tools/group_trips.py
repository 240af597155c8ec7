#!/usr/bin/env python3
"""Diagnostic: what a wavefront of wfa_group_kernel EXECUTES per score step -- trips of the k-loop and of extend's while loop -- counted by
a -DAIM_GROUP_COUNT_TRIPS=1 build (python -m aim_amd.build --variant trips --flags "-DAIM_GROUP_COUNT_TRIPS=1";
AIM_LIB=build_ab/lib_trips.so python tools/group_trips.py [l e n]).  Score-only; the counts ride in the result of every wavefront's first pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aim_amd import engine
shapes = [(int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]))] if len(sys.argv) > 3 else [(1000, 0.05, 65536), (100, 0.05, 1 << 18), (250, 0.05, 65536)]
for l, e, n in shapes:
    ms, rs = engine.launcher_sizes("wfa", l, e)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    params = engine.make_params("wfa", ms, rs, reduce=True)
    res, _ = engine.align(params, req, pat, txt, check=False)
    kt, et, st = (res[f].astype(np.float64) for f in ("max_operations", "begin_offset", "end_offset"))
    waves = (st > 0).sum()
    print("l=%d e=%g n=%d: %d wavefronts (%.1f pairs each), mean score %.1f, score steps per wavefront %.1f, k-trips per step %.2f, extend trips per "
          "k-trip %.2f, extend trips per step %.2f (one per k-trip would be %.2f)" % (l, e, n, waves, n / waves, res["score"].mean(), st.sum() / waves,
          kt.sum() / st.sum(), et.sum() / kt.sum(), et.sum() / st.sum(), kt.sum() / st.sum()))
