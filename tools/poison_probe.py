#!/usr/bin/env python3
"""Diagnostic: do results depend on scratch memory a launch did not write? Runs random configurations through the set API
(as the host CLI does: several pushes/launches on one configured set) with the scratch pre-filled with 0x00, 0x7f and 0xff
(AIM_DEBUG_POISON_SCRATCH) and requires identical results and ops in [begin, end); also checks them against the oracle.

    python tools/poison_probe.py [--seconds 120] [--seed 1] [--focus dp|all]"""
import argparse, json, os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import capi, engine
from oracle import oracle
import ctypes as C
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--focus", choices=["dp", "dp2", "all"], default="dp")
a = ap.parse_args()
rng = random.Random(a.seed)
lib = capi.load()

def run(params, batches, poison):
    os.environ["AIM_DEBUG_POISON_SCRATCH"] = str(poison)
    outs = []
    with engine.DeviceSet(1) as s:
        s.configure(params, max(len(b[0]) for b in batches))
        for req, pat, txt in batches:                       # several launches on one set, like the host's partitions
            s.push(0, req, pat, txt); s.launch()
            res, ops = s.pull(0, check=False)        # pairs the reference would abort on come back with a status
            outs.append((res.copy(), None if ops is None else ops.copy()))
    return outs

def canon(res, ops):
    key = [res[f].tobytes() for f in ("score", "status", "begin_offset", "end_offset", "max_operations")]
    if ops is not None:
        ok = res["status"] == 0
        key.append(b"".join(ops[i, int(res["begin_offset"][i]):int(res["end_offset"][i])].tobytes() for i in np.nonzero(ok)[0]))
    return key

t0, cases, hits = time.time(), 0, 0
while time.time() - t0 < a.seconds:
    algo = rng.choice(["nw", "nw", "swg"]) if a.focus in ("dp", "dp2") else rng.choice(["wfa", "nw", "swg"])
    l = rng.choice([700, 700, 1000]) if a.focus == "dp2" else rng.choice([100, 300, 700, 700, 1000]); e = rng.choice([0.02, 0.05, 0.1])
    ms, rs = engine.launcher_sizes(algo, l, e)
    kw = dict(backtrace=True)
    if algo == "wfa": kw["reduce"] = rng.random() < 0.6
    params = engine.make_params(algo, ms, rs, **kw)
    npd = rng.choice([8, 16, 32, 64])
    sizes = [npd] * rng.choice([1, 2, 3]) + ([rng.randint(1, npd)] if rng.random() < 0.5 else [])
    batches = [engine.gen_pairs(rng.randint(1, 1 << 30), 0, k, l, e, rs) for k in sizes]
    ref = None
    case = dict(algo=algo, l=l, e=e, read_size=rs, max_score=ms, sizes=sizes, kernel=lib.aim_kernel_name(C.byref(params)).decode())
    for poison in (0, 0x7f, 0xff):
        outs = run(params, batches, poison)
        key = [canon(r, o) for r, o in outs]
        if ref is None: ref = (key, outs)
        elif key != ref[0]:
            for bi, (k1, k2) in enumerate(zip(ref[0], key)):
                if k1 != k2:
                    r0, r1 = ref[1][bi][0], outs[bi][0]
                    bad = [i for i in range(len(r0)) if any(r0[f][i] != r1[f][i] for f in ("score", "status", "begin_offset", "end_offset"))]
                    print(json.dumps(dict(case, poison=poison, batch=bi, differing_pairs=bad[:8], note="results depend on unwritten scratch")), flush=True)
                    req, pat, txt = batches[bi]
                    op = oracle.params(algo, ms, rs, match=params.match, mismatch=params.mismatch, gap_o=params.gap_o, gap_e=params.gap_e, gap=params.gap_i,
                                       backtrace=True, reduce=bool(params.flags & capi.FLAG_REDUCE), swg_cell_bytes=0)
                    ores, oops, _ = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=16)
                    fl = ("score", "status", "begin_offset", "end_offset")
                    for i in bad[:4]:
                        print("  pair", i, "plen/tlen", int(req["pattern_len"][i]), int(req["text_len"][i]), "| poison 0:", {f: int(r0[f][i]) for f in fl},
                              "| poison %d:" % poison, {f: int(r1[f][i]) for f in fl}, "| oracle:", {f: int(ores[f][i]) for f in fl}, flush=True)
                    # the same batch alone on a fresh set, each poison twice: is it the poison or the launch history?
                    for pz in (0, 255, 0, 255):
                        o = run(params, [batches[bi]], pz)[0][0]
                        print("  alone, poison", pz, [{f: int(o[f][i]) for f in fl} for i in bad[:2]], flush=True)
            hits += 1
            if a.focus != "dp2": sys.exit(1)
            break
    os.environ.pop("AIM_DEBUG_POISON_SCRATCH", None)
    cases += 1
print(json.dumps({"cases": cases, "launch_sets": cases * 3, "seconds": round(time.time() - t0, 1), "hits": hits, "lib": os.environ.get("AIM_LIB", "default")}), flush=True)
