"""GenASM: residency (AIM_GA_PER_CU) against read length, kernel ms (best of 3), same box.
    python tools/ga_sweep.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from aim_amd import engine


def run(l, e, n):
    rs = ((int(l * (1 + e)) + 8 + 7) // 8) * 8
    params = engine.make_params("genasm", 0, rs, backtrace=True)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(3):
            k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
        return best, s.plan_describe(0)


for l, n in ((100, 1 << 18), (200, 1 << 17), (300, 1 << 17), (500, 1 << 16), (1000, 1 << 16), (2000, 1 << 15), (5000, 1 << 14), (10000, 1 << 13)):
    for env in ({}, {"AIM_GA_PER_CU": "16"}, {"AIM_GA_PER_CU": "32"}):
        os.environ.pop("AIM_GA_PER_CU", None)
        os.environ.update(env)
        ms_, plan = run(l, 0.10, n)
        print("l=%-5d n=%-7d %-24s %.3f ms  %.3g pairs/s | grid=%s lds=%s" % (l, n, env, ms_, n / ms_ * 1e3, plan.split("grid=")[1].split()[0], plan.split("lds=")[1].split()[0]), flush=True)
