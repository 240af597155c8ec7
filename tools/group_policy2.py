"""Round-2 plan probe for wfa_group: default plan vs forced lanes-per-pair / row length, score-only, kernel ms (best of 3)."""
import os, sys, json
sys.path.insert(0, os.getcwd())
from aim_amd import engine
def run(l, e, n):
    ms, rs = engine.launcher_sizes("wfa", l, e)
    params = engine.make_params("wfa", ms, rs, reduce=True)
    req, pat, txt = engine.gen_pairs(42, 0, n, l, e, rs)
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        best = None
        for _ in range(3):
            k0 = s.timers()[1]; s.push(0, req, pat, txt); s.launch(); k = s.timers()[1] - k0
            best = k if best is None else min(best, k)
        return best, s.fallback_pairs(0), s.plan_describe(0)
cases = ((400, 0.10, 1 << 15), (250, 0.10, 1 << 16), (500, 0.05, 1 << 16), (1000, 0.02, 1 << 16), (250, 0.05, 1 << 18), (2000, 0.05, 1 << 14))
envs = [{}] + [{"AIM_GROUP_G": g, "AIM_GROUP_WLDS": w} for g in ("16", "32", "64") for w in ("0", "128")]
for l, e, n in cases:
    for env in envs:
        for k in ("AIM_GROUP_G", "AIM_GROUP_WLDS"): os.environ.pop(k, None)
        os.environ.update(env)
        try:
            ms_, fb, plan = run(l, e, n)
            print("l=%d e=%g %-44s %.3f ms fallback %d | G=%s" % (l, e, json.dumps(env), ms_, fb, plan.split(" G=")[1][:3]), flush=True)
        except Exception as ex:
            print("l=%d e=%g %-44s FAILED %s" % (l, e, json.dumps(env), str(ex)[:60]), flush=True)
