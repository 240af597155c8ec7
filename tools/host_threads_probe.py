#!/usr/bin/env python3
"""Where does the host CLI's time go as threads are added? Runs `host` on a text file of `copies` Mi pairs with the
output sent to /dev/null (no inode bound) for a sweep of --pack-threads / --format-threads and prints the CLI's own phase summary.

    python tools/host_threads_probe.py [copies=16] [--packed]
"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import engine
args = sys.argv[1:]
copies = int(args[0]) if args and args[0].isdigit() else 16
d = os.environ.get("AIM_SCALE_DIR", "/tmp")
l, err, unit = 100, 0.01, 1 << 20
ms, rs = engine.launcher_sizes("wfa", l, err)
base = os.path.join(d, "aim_unit_%d.seq" % unit)
big = os.path.join(d, "aim_scale_%dx.seq" % copies)
n = unit * copies
if not os.path.exists(big):
    if not os.path.exists(base):
        with open(base, "wb") as f:
            for i in range(0, unit, 1 << 16):
                req, pat, txt = engine.gen_pairs(42, i, 1 << 16, l, err, rs)
                f.write(engine.pairs_to_text(req, pat, txt))
    blob = open(base, "rb").read()
    with open(big, "wb") as out:
        for _ in range(copies):
            out.write(blob)
host = os.path.join(ROOT, "aim_amd", "host", "host")
common = [str(n), "--algo", "wfa", "--max-score", str(ms), "--read-size", str(rs), "--reduce"]
print("# cpus", os.cpu_count(), open("/proc/loadavg").read().strip(), flush=True)
os.system("lscpu | grep -i -E 'numa|socket|thread|model name' 1>&2")
for extra_name, extra in (("score", []), ("cigar", ["--backtrace"])):
    for pt, ft in ((8, 4), (16, 8), (32, 16), (64, 32), (96, 48), (128, 64), (170, 85)):
        best = None
        for rep in range(2):
            r = subprocess.run([host, big, "/dev/null"] + common + extra + ["--pack-threads", str(pt), "--format-threads", str(ft)] + [a for a in args[1:] if a.startswith("--")],
                               capture_output=True, text=True, cwd=d, env=dict(os.environ))
            if r.returncode:
                print("rc", r.returncode, r.stderr[-300:]); break
            m = re.search(r"parse\+pack ([\d.]+) ms \(line index ([\d.]+) ms\), wait ([\d.]+) ms, format\+write ([\d.]+) ms, loop ([\d.]+) ms, steady ([\d.eE+-]+) pairs/s \(in the loop: pack join ([\d.]+) ms, writer hand-over ([\d.]+) ms, submit ([\d.]+) ms\)", r.stdout)
            row = dict(zip(("parse_pack", "index", "wait", "fmt_write", "loop", "steady", "join", "wr", "submit"), map(float, m.groups())))
            if best is None or row["loop"] < best["loop"]:
                best = row
        print(json.dumps({"out": extra_name, "pack_threads": pt, "fmt_threads": ft, **best}), flush=True)
