#!/usr/bin/env python3
"""dp_strip_kernel's shapes (cells per lane K x wavefronts per pair) over READ_SIZE 2049 .. ~5500, score-only and with CIGAR, few and many pairs: kernel
time with AIM_STRIP_K forced to 16 / 20 / 24 / 32 (AIM_NO_DP_GROUP=1) next to the plan's own choice -- the data behind dp_strip_shape's cost rule (round 6).
Usage: python3 tools/strip_shape_sweep.py [out.txt]"""
import sys, os, json, io, contextlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_configs as bc

out = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/strip_shape_sweep.txt", "w")
LENGTHS = [int(x) for x in os.environ.get("SWEEP_LENGTHS", "2000,2400,2800,3500,4000,5000").split(",")]
NS = [int(x) for x in os.environ.get("SWEEP_NS", "1024,4096").split(",")]
for algo in ("nw", "swg"):
    for bt in (False, True):
        for l in LENGTHS:
            for n in NS:
                row = []
                for k in ("plan", "16", "20", "24", "32"):
                    os.environ.pop("AIM_STRIP_K", None); os.environ.pop("AIM_NO_DP_GROUP", None)
                    if k != "plan":
                        os.environ["AIM_STRIP_K"] = k
                        os.environ["AIM_NO_DP_GROUP"] = "1"
                    kw = dict(backtrace=True) if bt else {}
                    if algo == "swg": kw["swg_w16"] = True
                    buf = io.StringIO()
                    try:
                        with contextlib.redirect_stdout(buf):
                            bc.run("x", dict(algo=algo, l=l, e=0.05, n=n, kw=kw), reps=2)
                        d = json.loads(buf.getvalue().strip().split("\n")[-1])
                        shape = d["plan"].split()[0].replace("_kernel", "")
                        for tok in d["plan"].split():
                            if tok.startswith("block="): shape += " w%d" % (int(tok[6:]) // 64)
                            if tok.startswith("cells_per_lane="): shape += " k" + tok[15:]
                        row.append("%s: %s %.2f ms %.0f" % (k, shape, d["kernel_ms"], d["gcups"]))
                    except Exception as e:
                        row.append("%s: FAILED %r" % (k, e))
                line = "%-3s %-5s rs=%d n=%-5d | " % (algo, "cigar" if bt else "score", d.get("read_size", 0), n) + " | ".join(row)
                print(line, flush=True); out.write(line + "\n"); out.flush()
