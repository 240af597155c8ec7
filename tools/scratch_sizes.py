#!/usr/bin/env python3
"""Diagnostic: aim_scratch_bytes of every tools/bench_configs.py configuration (+ the headline) under the current budget."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine
from tools.bench_configs import CONFIGS
lib = capi.load()
rows = dict(CONFIGS); rows["headline_wfa_l100_e1_score_4M"] = dict(algo="wfa", l=100, e=0.01, n=4 << 20, kw=dict(reduce=True))
for name, c in rows.items():
    ms, rs = engine.launcher_sizes(c["algo"], c["l"], c["e"])
    p = engine.make_params(c["algo"], ms, rs, **c["kw"])
    print("%-30s %-16s n=%-8d scratch %10.3f MB" % (name, lib.aim_kernel_name(C.byref(p)).decode(), c["n"], lib.aim_scratch_bytes(C.byref(p), c["n"]) / 1e6))
