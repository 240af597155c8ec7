#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of a -DAIM_DPW_STAMPS -DAIM_DPW_DIAG_NO_TRACEBACK build of dp_wave_kernel on
config 4 (thread 0 of each workgroup; dumped into the pair's ops row)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
n, l, e = 128, 10000, 0.01
ms, rs = engine.launcher_sizes("swg", l, e)
params = engine.make_params("swg", ms, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
with engine.DeviceSet(1) as s:
    s.configure(params, n)
    s.push(0, req, pat, txt); s.launch()
    k = s.timers()[1]
    res, ops = s.pull(0)
st = np.ascontiguousarray(ops[:, :64]).view(np.uint64).astype(np.float64)      # [pair][8]
names = ["row-start barrier", "row reads + A/I/G", "wave scan", "carry barrier", "carry reads + M/D", "pack + stores", "tail barrier", "tail cell / loop"]
tail = req["pattern_len"].astype(int) > req["text_len"].astype(int)
rows = req["text_len"].astype(np.float64)
print("kernel %.2f ms" % k)
for label, m in (("pairs with a tail (%d)" % tail.sum(), tail), ("pairs without (%d)" % (~tail).sum(), ~tail)):
    tot = st[m].sum(axis=1).mean()
    print("%s: %.0f ticks per pair, %.0f per row" % (label, tot, (st[m].sum(axis=1) / rows[m]).mean()))
    for i, nm in enumerate(names):
        print("   %-20s %7.0f ticks/row %5.1f %%" % (nm, (st[m][:, i] / rows[m]).mean(), 100 * st[m][:, i].mean() / tot))
