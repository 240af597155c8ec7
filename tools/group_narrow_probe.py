import os, sys, json, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from aim_amd import capi, engine
def run(l, e, n, bt):
    ms, rs = engine.launcher_sizes("wfa", l, e)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=bt)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(3):
            k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
        fb = s.fallback_pairs(0); plan = s.plan_describe(0)
        res, ops = s.pull(0)
    return best, fb, plan, res
for env in ({}, {"AIM_GROUP_WLDS": "0"}, {"AIM_GROUP_G": "16"}, {"AIM_GROUP_G": "32"}, {"AIM_GROUP_G": "64"}, {"AIM_GROUP_WLDS": "64"}, {"AIM_GROUP_WLDS": "256"}):
    for k in ("AIM_GROUP_WLDS", "AIM_GROUP_G"): os.environ.pop(k, None)
    os.environ.update(env)
    for l, e, n, bt in ((1000, 0.05, 65536, False), (1000, 0.05, 65536, True)):
        ms_, fb, plan, res = run(l, e, n, bt)
        print(json.dumps(env), "l=%d bt=%d: %.3f ms  %.4g pairs/s  fallback %d  mean score %.2f | %s" % (l, bt, ms_, n / ms_ * 1e3, fb, res["score"].mean(), plan.split("budget")[1][-40:]))
