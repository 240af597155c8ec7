import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from aim_amd import engine
def run(l, e, n, bt, red=True):
    ms, rs = engine.launcher_sizes("wfa", l, e)
    params = engine.make_params("wfa", ms, rs, reduce=red, backtrace=bt)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(3):
            k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
        fb = s.fallback_pairs(0); plan = s.plan_describe(0)
    return best, fb, plan
cases = ((100, 0.05, 1 << 20, False), (100, 0.10, 1 << 19, False), (250, 0.05, 1 << 18, False), (150, 0.02, 1 << 20, False), (100, 0.05, 1 << 19, True), (500, 0.05, 1 << 16, False), (400, 0.10, 1 << 15, False))
for env in ({}, {"AIM_GROUP_WLDS": "64"}, {"AIM_GROUP_WLDS": "32"}, {"AIM_GROUP_WLDS": "128"}):
    os.environ.pop("AIM_GROUP_WLDS", None); os.environ.update(env)
    for l, e, n, bt in cases:
        ms_, fb, plan = run(l, e, n, bt)
        print(json.dumps(env), "l=%d e=%g bt=%d: %.3f ms %.4g pairs/s fallback %d (%.2f%%) | %s" % (l, e, bt, ms_, n / ms_ * 1e3, fb, 100.0 * fb / n, plan.split(" G=")[1][:12]))
