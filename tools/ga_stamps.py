#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of a -DAIM_GA_STAMPS=1 build of genasm_wave_kernel (BASELINE config 5 shape)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rs = ((int(L * (1 + E)) + 8 + 7) // 8) * 8
params = engine.make_params("genasm", 0, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(42, 0, n, L, E, rs)
res, ops = engine.align(params, req, pat, txt)
st = np.ascontiguousarray(ops[:, :64]).view(np.uint64).astype(np.float64)   # [n][8]
windows = L / 40.0
names = ["window chars (HBM)", "pattern masks", "DC fast path (<= 15 edits)", "DC full-width path", "traceback", "ops stores"]
tot = st[:, :6].sum(axis=1).mean()
print("ticks per pair %.0f, per window (~%d windows) %.0f" % (tot, windows, tot / windows))
for i, nm in enumerate(names):
    print("%-28s %8.0f ticks/window %5.1f%%" % (nm, st[:, i].mean() / windows, 100 * st[:, i].mean() / tot))
