#!/usr/bin/env python3
"""Wall-clock of the host CLI on a synthetic text file (parse + H2D + kernel + D2H + write), per phase.
    python tools/cli_e2e.py [pairs] [--backtrace]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1 << 20
bt = "--backtrace" in sys.argv
ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
path = "/tmp/aim_pairs_%d.seq" % n
if not os.path.exists(path):
    t0 = time.time()
    with open(path, "wb") as f:
        step = 1 << 16
        for i in range(0, n, step):
            req, pat, txt = engine.gen_pairs(42, i, min(step, n - i), 100, 0.01, rs)
            f.write(engine.pairs_to_text(req, pat, txt))
    print("generated %s in %.1f s" % (path, time.time() - t0))
host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aim_amd", "host", "host")
cmd = [host, path, "/tmp/aim_out.txt", str(n), "--algo", "wfa", "--max-score", str(ms), "--read-size", str(rs), "--reduce"] + (["--backtrace"] if bt else [])
for rep in range(2):
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
    dt = time.time() - t0
    print("run %d: %.3f s wall -> %.2e pairs/s  rc=%d" % (rep, dt, n / dt, r.returncode))
print(r.stdout[-600:])
print(subprocess.run(["md5sum", "/tmp/aim_out.txt"], capture_output=True, text=True).stdout.strip())
