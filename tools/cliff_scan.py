#!/usr/bin/env python3
"""Coarse scan for performance cliffs: every algorithm x read length x error rate x {score-only, CIGAR} at the launchers' own sizes, one small launch each (cells bounded, so that
even a one-lane literal path finishes in seconds); prints kernel, plan shape, ms and GCUPS (WFA: pairs/s) and marks rows far below their neighbours. Round 6: the NW literal
cliff at READ_SIZE >= 3 998 (4 GCUPS) was found by accident -- this is the systematic version. Usage: python3 tools/cliff_scan.py [out.txt]"""
import sys, os, json, io, contextlib, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_configs as bc

out = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cliff_scan.txt", "w")
LENGTHS = [int(x) for x in os.environ.get("SCAN_LENGTHS", "50,100,150,200,300,500,800,1000,1500,2000,3000,4000,6000,8000,10000,15000").split(",")]
ERRS = [float(x) for x in os.environ.get("SCAN_ERRS", "0.01,0.05,0.10").split(",")]
BUDGET = float(os.environ.get("SCAN_CELLS", "4e9"))      # cells per launch (DP); WFA: pairs bounded the same way
t_end = time.time() + float(os.environ.get("SCAN_SECONDS", "900"))
for algo, kw0 in (("nw", {}), ("swg", dict(swg_w16=True)), ("swg", {}), ("wfa", dict(reduce=True))):
    for bt in (False, True):
        for l in LENGTHS:
            for e in ERRS:
                if time.time() > t_end: break
                n = int(min(1 << 20, max(64, BUDGET / (l * l))))
                if l >= 3000: n = min(n, 256)
                kw = dict(kw0)
                if bt: kw["backtrace"] = True
                buf = io.StringIO()
                tag = "%s%s l=%d e=%g %s n=%d" % (algo, "_w16" if kw0.get("swg_w16") else "", l, e, "cigar" if bt else "score", n)
                try:
                    t0 = time.time()
                    with contextlib.redirect_stdout(buf):
                        bc.run("x", dict(algo=algo, l=l, e=e, n=n, kw=kw), reps=1)
                    d = json.loads(buf.getvalue().strip().split("\n")[-1])
                    shape = " ".join(t for t in d["plan"].split() if t.startswith(("block=", "cells_per_lane=", "lanes_per_pair=", "wavefronts_per_pair=")))
                    line = "%-44s %-24s %-48s %10.3f ms %12.1f GCUPS %12.4g pairs/s todo=%s wall=%.1fs" % (tag, d["kernel"], shape, d["kernel_ms"], d["gcups"], d["pairs_per_s"], d["todo_pairs"], time.time() - t0)
                except Exception as ex:
                    line = "%-44s FAILED %r" % (tag, ex)
                print(line, flush=True); out.write(line + "\n"); out.flush()
