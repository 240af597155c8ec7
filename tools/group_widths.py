import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from aim_amd import engine
for l, e, n in ((1000, 0.05, 4096), (100, 0.05, 8192), (250, 0.05, 4096)):
    ms, rs = engine.launcher_sizes("wfa", l, e)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    res, _ = engine.align(engine.make_params("wfa", ms, rs, reduce=True), req, pat, txt, check=False)
    steps = res["score"].astype(np.float64)
    print("l=%d e=%g: mean score %.1f, mean width per computed step %.1f, steps>32: %.1f%%, steps>64: %.1f%%" % (
        l, e, steps.mean(), (res["max_operations"] / np.maximum(1, steps)).mean(), 100 * (res["begin_offset"] / np.maximum(1, steps)).mean(),
        100 * (res["end_offset"] / np.maximum(1, steps)).mean()))
