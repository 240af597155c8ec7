#!/usr/bin/env python3
"""Diagnostic: dp_wave kernel time against wavefronts per pair (AIM_DPW_NW) at a given length / pair count.
usage: tools/dpw_nw_probe.py <algo nw|swg> <length> <error> <n_pairs> <nw> [<nw> ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
algo, l, e, n = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
ms, rs = engine.launcher_sizes(algo, l, e)
params = engine.make_params(algo, ms, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
row = {"algo": algo, "l": l, "e": e, "n": n, "read_size": rs}
for nw in sys.argv[5:]:
    if nw == "default": os.environ.pop("AIM_DPW_NW", None)
    else: os.environ["AIM_DPW_NW"] = nw
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(3):
            k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
    row["nw_" + nw] = round(best, 3)
print(json.dumps(row))
