#!/usr/bin/env python3
"""Diagnostic: does the sequential tail phase (pairs with plen > tlen) set config 4's kernel time?
Times the standard 128 synthetic pairs, the same pairs with pattern/text swapped wherever plen > tlen (no tails at
all), and swapped the other way (every pair has a tail); prints the tail-length distribution."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine

n, l, e = 128, 10000, 0.01
ms, rs = engine.launcher_sizes("swg", l, e)
params = engine.make_params("swg", ms, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)

def timed(req, pat, txt):
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(3):
            k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
    return best

def swapped(cond):
    r, p, t = req.copy(), pat.copy(), txt.copy()
    for i in np.nonzero(cond)[0]:
        r["pattern_len"][i], r["text_len"][i] = req["text_len"][i], req["pattern_len"][i]
        p[i], t[i] = txt[i], pat[i]
    return r, p, t

d = req["pattern_len"].astype(int) - req["text_len"].astype(int)
print(json.dumps({"plen_minus_tlen": {"min": int(d.min()), "max": int(d.max()), "mean": float(d.mean()), "pairs_with_tail": int((d > 0).sum())}}))
print(json.dumps({"standard_ms": timed(req, pat, txt)}))
print(json.dumps({"no_tails_ms": timed(*swapped(d > 0))}))
print(json.dumps({"all_tails_ms": timed(*swapped(d < 0))}))
