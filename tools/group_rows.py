"""wfa_group: entries per LDS ring row (AIM_GROUP_WLDS) against kernel time and the pairs that leave for the general kernel
(to-do list), score-only and with CIGAR, on the shapes that run in narrow mode. Kernel ms = best of 3; same box.
    python tools/group_rows.py [W ...]"""
import os, sys, json
sys.path.insert(0, os.getcwd())
from aim_amd import engine


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
        return best, s.fallback_pairs(0), s.plan_describe(0)


cases = ((1000, 0.05, 1 << 16), (400, 0.10, 1 << 15), (250, 0.10, 1 << 16), (500, 0.05, 1 << 16), (1000, 0.02, 1 << 16), (2000, 0.05, 1 << 14))
widths = sys.argv[1:] or ["128", "112", "96", "80", "64"]
for l, e, n in cases:
    for bt in (False, True):
        for w in widths:
            os.environ["AIM_GROUP_WLDS"] = w
            try:
                ms_, fb, plan = run(l, e, n, bt)
                print("l=%d e=%g %s W=%-4s %.3f ms  to-do %d of %d | grid=%s G=%s" % (l, e, "cigar" if bt else "score", w, ms_, fb, n, plan.split("grid=")[1].split()[0], plan.split(" G=")[1][:3]), flush=True)
            except Exception as ex:
                print("l=%d e=%g W=%s FAILED %s" % (l, e, w, str(ex)[:80]), flush=True)
