#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of a -DAIM_GA_STAMPS=1 build when every window takes the 64-level path (unrelated texts)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
L, n = 20000, 256
rs = ((int(L * 1.1) + 8 + 7) // 8) * 8
params = engine.make_params("genasm", 0, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(42, 0, n, L, 0.1, rs)
rng = np.random.RandomState(1)
for i in range(n):
    tl = int(req["text_len"][i]); txt[i, :tl] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=tl)
res, ops = engine.align(params, req, pat, txt)
st = np.ascontiguousarray(ops[:, :64]).view(np.uint64).astype(np.float64)
windows = L / 40.0
names = ["window chars (HBM)", "pattern masks", "DC fast path (<= 15 edits)", "DC full-width path", "traceback", "ops stores"]
tot = st[:, :6].sum(axis=1).mean()
print("unrelated texts: ticks per pair %.0f, per window (~%d windows) %.0f; mean score %.0f" % (tot, windows, tot / windows, res["score"].mean()))
for i, nm in enumerate(names):
    print("%-28s %8.0f ticks/window %5.1f%%" % (nm, st[:, i].mean() / windows, 100 * st[:, i].mean() / tot))
