#!/usr/bin/env python3
"""Diagnostic: run the SAME launch many times and compare every result with the oracle's -- a rate detector for
nondeterminism. usage: tools/repeat_launch.py <algo> <length> <error> <n_pairs> <launches> [seed]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import capi, engine
from oracle import oracle
import ctypes as C
algo, l, e, n, reps = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 5
ms, rs = engine.launcher_sizes(algo, l, e)
params = engine.make_params(algo, ms, rs, backtrace=True)
req, pat, txt = engine.gen_pairs(seed, 0, n, l, e, rs)
op = oracle.params(algo, ms, rs, match=0, mismatch=3, gap_o=4, gap_e=1, gap=4, backtrace=True, reduce=False, swg_cell_bytes=0)
ores, oops, _ = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=32)
fl = ("score", "status", "begin_offset", "end_offset")
bad_launches, examples = 0, []
t0 = time.time()
with engine.DeviceSet(1) as s:
    s.configure(params, n)
    s.push(0, req, pat, txt)
    for it in range(reps):
        s.launch()
        res, ops = s.pull(0, check=False)
        bad = [i for i in range(n) if any(res[f][i] != ores[f][i] for f in fl)
               or (res["status"][i] == 0 and not np.array_equal(ops[i, res["begin_offset"][i]:res["end_offset"][i]], oops[i, ores["begin_offset"][i]:ores["end_offset"][i]]))]
        if bad:
            bad_launches += 1
            if len(examples) < 4:
                i = bad[0]
                examples.append(dict(launch=it, pairs=bad[:6], plen=int(req["pattern_len"][i]), tlen=int(req["text_len"][i]),
                                     hip={f: int(res[f][i]) for f in fl}, oracle={f: int(ores[f][i]) for f in fl}))
print(json.dumps(dict(algo=algo, l=l, e=e, n=n, kernel=capi.load().aim_kernel_name(C.byref(params)).decode(), env={k: v for k, v in os.environ.items() if k.startswith("AIM_")},
                      launches=reps, bad_launches=bad_launches, seconds=round(time.time() - t0, 1), examples=examples)), flush=True)
