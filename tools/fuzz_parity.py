#!/usr/bin/env python3
"""Randomised differential run: HIP path (through the C-ABI) against the CPU oracle, bit-exact, over random algorithms,
lengths, error rates, penalties, flags, pair counts and forced plans. The oracle is used here as the checker only.

    python tools/fuzz_parity.py [--seconds 120] [--seed 1] [--max-cells 4e8]

Prints one line per case and a summary; exits 1 on the first mismatch (the case's parameters are printed, so it can be
turned into a fixed test)."""
import argparse, json, os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from aim_amd import capi, engine
from oracle import oracle
import ctypes as C


def oracle_params(params, algo):
    bt = bool(params.flags & capi.FLAG_BACKTRACE)
    red = bool(params.flags & capi.FLAG_REDUCE)
    cellb = 2 if (params.flags & capi.FLAG_SWG_W16) else 0
    return oracle.params(algo, params.max_score, params.read_size, match=params.match, mismatch=params.mismatch,
                         gap_o=params.gap_o, gap_e=params.gap_e, gap=params.gap_i, backtrace=bt, reduce=red, swg_cell_bytes=cellb)


def compare(algo, params, req, pat, txt):
    res, ops = engine.align(params, req, pat, txt, check=False)
    ores, oops, _ = oracle.align_batch(oracle_params(params, algo), req["pattern_len"], req["text_len"], pat, txt, nthreads=64)
    for f in ("score", "max_operations", "end_offset", "status"):
        bad = np.nonzero(res[f] != ores[f])[0]
        if bad.size: return "%s differs at pair %d: hip %d oracle %d" % (f, bad[0], res[f][bad[0]], ores[f][bad[0]])
    if params.flags & capi.FLAG_BACKTRACE:
        ok = res["status"] == 0
        bad = np.nonzero((res["begin_offset"] != ores["begin_offset"]) & ok)[0]
        if bad.size: return "begin_offset differs at pair %d" % bad[0]
        for i in np.nonzero(ok)[0]:
            b, e = int(res["begin_offset"][i]), int(res["end_offset"][i])
            if not np.array_equal(ops[i, b:e], oops[i, b:e]): return "ops differ at pair %d" % i
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=float, default=4e8, help="bound on n * l * l per case (oracle time)")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    lib = capi.load()
    t0, cases, kernels = time.time(), 0, {}
    while time.time() - t0 < a.seconds:
        algo = rng.choice(["wfa", "wfa", "wfa", "nw", "swg"])
        l = rng.choice([20, 33, 64, 100, 100, 150, 250, 300, 400, 700, 1000, 1500, 2500])
        e = rng.choice([0.0, 0.01, 0.02, 0.05, 0.10, 0.15])
        cost = {}
        if rng.random() < 0.5:
            if algo == "wfa": cost = dict(mismatch=rng.randint(1, 6), gap_o=rng.randint(1, 6), gap_e=rng.randint(1, 3))
            elif algo == "swg": cost = dict(mismatch=rng.randint(1, 6), gap_o=rng.randint(1, 6), gap_e=rng.randint(1, 3))
            else: cost = dict(mismatch=rng.randint(1, 6), gap=rng.randint(1, 6))
        try:
            ms, rs = engine.launcher_sizes(algo, l, e, **cost)
        except Exception:
            continue
        if algo == "wfa" and rng.random() < 0.2: ms = max(1, ms // 2)          # some pairs exceed MAX_SCORE
        n = int(min(rng.choice([1, 7, 64, 65, 300, 1000, 5000]), max(1, a.max_cells // (l * l))))
        kw = dict(backtrace=rng.random() < 0.6, **cost)
        if algo == "wfa": kw["reduce"] = rng.random() < 0.6
        if algo == "swg" and rng.random() < 0.3: kw["swg_w16"] = True
        try:
            params = engine.make_params(algo, ms, rs, **kw)
        except Exception:
            continue
        env = {}
        r = rng.random()
        if algo == "wfa" and r < 0.25: env["AIM_GROUP_G"] = rng.choice(["1", "2", "4", "8", "16", "64"])
        if algo == "wfa" and 0.25 <= r < 0.35: env["AIM_FORCE_WAVE"] = "1"
        if algo != "wfa" and r < 0.4: env["AIM_DPW_NW"] = rng.choice(["1", "2", "4"])
        if algo != "wfa" and 0.4 <= r < 0.5: env["AIM_FORCE_DPWAVE"] = "1"
        for k in ("AIM_GROUP_G", "AIM_FORCE_WAVE", "AIM_DPW_NW", "AIM_FORCE_DPWAVE"): os.environ.pop(k, None)
        os.environ.update(env)
        req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
        if n > 3 and rng.random() < 0.3: pat[rng.randrange(n), rng.randrange(max(1, l // 2))] = ord("N")   # non-ACGT byte
        kn = lib.aim_kernel_name(C.byref(params)).decode()
        case = dict(algo=algo, l=l, e=e, n=n, max_score=ms, read_size=rs, kernel=kn, env=env, **{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
        try:
            err = compare(algo, params, req, pat, txt)
        except Exception as ex:
            err = "exception: %r" % (ex,)
        cases += 1
        kernels[kn] = kernels.get(kn, 0) + 1
        print(json.dumps(dict(case, ok=err is None)), flush=True)
        if err:
            print("MISMATCH:", err, flush=True)
            return 1
    print(json.dumps({"cases": cases, "seconds": round(time.time() - t0, 1), "kernels": kernels, "all_ok": True}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
