#!/usr/bin/env python3
"""Diagnostic: wfa_group_kernel lanes-per-pair policy. Times a spread of (l, e) WFA-adaptive configurations under the
default plan and with AIM_GROUP_G forced, kernel timers through the set calls (best of 3), result digests compared."""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine

CASES = [(100, 0.10, 1 << 19), (250, 0.05, 1 << 18), (250, 0.10, 1 << 17), (400, 0.10, 1 << 16), (500, 0.05, 1 << 17),
         (1000, 0.02, 1 << 16), (1000, 0.05, 1 << 16)]
bt = len(sys.argv) > 1 and sys.argv[1] == "cigar"
for l, e, n in CASES:
    ms, rs = engine.launcher_sizes("wfa", l, e)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=bt)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    row = {"l": l, "e": e, "n": n, "max_score": ms, "cigar": bt}
    digests = set()
    for label, g in (("default", None), ("G16", "16"), ("G64", "64")):
        if g is None: os.environ.pop("AIM_GROUP_G", None)
        else: os.environ["AIM_GROUP_G"] = g
        with engine.DeviceSet(1) as s:
            s.configure(params, n)
            best = None
            for _ in range(3):
                k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
                best = k if best is None else min(best, k)
            res, ops = s.pull(0)
        h = hashlib.sha256(res["score"].tobytes() + (ops.tobytes() if ops is not None else b"")).hexdigest()[:12]
        digests.add(h)
        row[label] = round(n / (best * 1e-3), 1)
    os.environ.pop("AIM_GROUP_G", None)
    row["identical_results"] = len(digests) == 1
    print(json.dumps(row), flush=True)
