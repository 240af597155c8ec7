#!/usr/bin/env python3
"""Sweep of AIM_REG_PHASE (dp_reg.hpp: reg_phase_shift -- how late a SIMD's odd wavefront starts) for nw_reg / swg_reg with CIGAR: kernel ms per 1 Mi pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aim_amd import engine
for algo, l, e, n in (("nw", 100, 0.01, 1 << 20), ("swg", 100, 0.01, 1 << 20), ("nw", 100, 0.05, 1 << 20), ("nw", 150, 0.01, 1 << 20), ("nw", 100, 0.01, 1 << 22)):
    ms, rs = engine.launcher_sizes(algo, l, e)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    cells = float((req["pattern_len"].astype(np.int64) * req["text_len"]).sum())
    out = []
    for ph in ("0", "6", "12", "22", "35", "50", "-1"):
        os.environ["AIM_REG_PHASE"] = ph
        with engine.DeviceSet(1) as s:
            s.configure(engine.make_params(algo, ms, rs, backtrace=True), n)
            best = None
            for _ in range(3):
                k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
                best = k if best is None else min(best, k)
        out.append("%s: %.3f ms (%.0f)" % (ph, best, cells / best / 1e6))
    print("%s l=%d e=%g n=%d  " % (algo, l, e, n) + "; ".join(out), flush=True)
