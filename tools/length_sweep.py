#!/usr/bin/env python3
"""NW / SWG kernel timers over read lengths 150 .. 2000 (e = 2 %, score-only and with CIGAR): where one kernel class hands over to the next
(register kernels <= READ_SIZE 176, dp_group_kernel <= 1024, dp_strip_kernel above). Round 5: the sweep that showed the valley at l = 180 .. 700
(profiles/r05/length_sweep_before.txt) that dp_group.hpp fills (length_sweep_after.txt). Usage: python3 tools/length_sweep.py [out.txt]
(SWEEP_E=0.05: another error rate; SWEEP_CIGAR=1: the rows with CIGAR only; SWEEP_LENGTHS=300,700: a subset -- for A/Bs with AIM_LIB)"""
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_configs as bc
out = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/length_sweep.txt", "w")
import io, contextlib
E = float(os.environ.get("SWEEP_E", "0.02"))
LENGTHS = [int(x) for x in os.environ["SWEEP_LENGTHS"].split(",")] if os.environ.get("SWEEP_LENGTHS") else (150, 180, 200, 250, 300, 320, 400, 500, 700, 1000, 1200, 1450, 1500, 1700, 2000)
for algo in ("nw", "swg"):
    for bt in ((True,) if os.environ.get("SWEEP_CIGAR") else (False, True)):
        for l in LENGTHS:
            n = max(1024, int(2.5e10 / (l * l) / (4 if bt else 1)) // 1024 * 1024)
            n = min(n, 1 << 20)
            kw = dict(backtrace=True) if bt else {}
            if algo == "swg": kw["swg_w16"] = True   # (int16 cells: with the launcher's MAX_SCORE < 127 the reference's int8 cells wrap on these lengths)
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    bc.run("%s_l%d_%s" % (algo, l, "cigar" if bt else "score"), dict(algo=algo, l=l, e=E, n=n, kw=kw))
                d = json.loads(buf.getvalue().strip().split("\n")[-1])
                line = "%-18s %-16s rs=%5d n=%7d ms=%8.3f gcups=%8.1f todo=%s" % (d["config"], d["kernel"], d["read_size"], d["pairs"], d["kernel_ms"], d["gcups"], d["todo_pairs"])
            except Exception as e:
                line = "%s l=%d bt=%s FAILED %r" % (algo, l, bt, e)
            print(line, flush=True); out.write(line + "\n"); out.flush()
