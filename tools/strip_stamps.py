#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of a -DAIM_STRIP_STAMPS=1 build of dp_strip_kernel (BASELINE config 4 shape), per wavefront,
averaged over the pairs with / without a tail (plen > tlen: the boundary cell of a row then comes from the last strip of the row before).
    AIM_LIB=build_ab/lib_stamps.so python tools/strip_stamps.py [l=10000] [e=0.01] [n=256]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
E = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ms, rs = engine.launcher_sizes("swg", L, E)
params = engine.make_params("swg", ms, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(42, 0, n, L, E, rs)
res, ops = engine.align(params, req, pat, txt)
nw = 12
st = np.ascontiguousarray(ops[:, 64:64 + nw * 64]).view(np.uint64).astype(np.float64).reshape(n, nw, 8)
names = ["0 diagonal input (mailbox D)", "1 pre-carry", "2 boundary cell B(h)", "3 totals on the left (mailbox C)",
         "4 post-carry + post D", "5 tail cell / picks", "6 wave scan + post C (+ ring wait)", "7 table stores + back-edge"]
tail = req["pattern_len"] > req["text_len"]
rows = req["text_len"].astype(np.float64)
for label, sel in (("pairs with a tail (%d)" % tail.sum(), tail), ("pairs without (%d)" % (~tail).sum(), ~tail)):
    if not sel.any():
        continue
    per_row = st[sel] / rows[sel][:, None, None]          # ticks per row (on this chip a tick of s_memtime is ~0.5 ns: 5 300 ticks per row at 26.8 ms per 10 112 rows)
    used = per_row.sum(axis=2).mean(axis=0) > 0
    print(label, "-- s_memtime ticks per row (~0.5 ns each), by wavefront (columns) and phase (rows)")
    used &= per_row.sum(axis=2).mean(axis=0) < 1e9      # (slots of wavefronts the shape does not have hold CIGAR bytes)
    print("%-36s" % "" + "".join("%8s" % ("w%d" % w) for w in range(nw) if used[w]))
    for i, nm in enumerate(names):
        print("%-36s" % nm + "".join("%8.2f" % per_row[:, w, i].mean() for w in range(nw) if used[w]))
    print("%-36s" % "sum" + "".join("%8.2f" % per_row[:, w, :].sum(axis=1).mean() for w in range(nw) if used[w]))
