#!/usr/bin/env python3
"""The host CLI at scale: wall clock and phase times of `aim_amd/host/host` on a large input file --
text and packed input, score-only and CIGAR. The input is a seeded 1 Mi-pair file replicated `copies` times (content repeats,
pair indices do not), written under /tmp (or $AIM_SCALE_DIR); one JSON line per run.

    python tools/cli_scale.py [copies=64] [--threads T] [--only text|packed] [--keep]
"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import engine

args = sys.argv[1:]
copies = int(args[0]) if args and args[0].isdigit() else 64
threads = args[args.index("--threads") + 1] if "--threads" in args else None
only = args[args.index("--only") + 1] if "--only" in args else None
d = os.environ.get("AIM_SCALE_DIR", "/tmp")
l, err, unit = 100, 0.01, 1 << 20
ms, rs = engine.launcher_sizes("wfa", l, err)
base = os.path.join(d, "aim_unit_%d.seq" % unit)
big = os.path.join(d, "aim_scale_%dx.seq" % copies)
n = unit * copies
if not os.path.exists(big):
    t0 = time.time()
    if not os.path.exists(base):
        with open(base, "wb") as f:
            for i in range(0, unit, 1 << 16):
                req, pat, txt = engine.gen_pairs(42, i, 1 << 16, l, err, rs)
                f.write(engine.pairs_to_text(req, pat, txt))
    with open(big, "wb") as out:
        blob = open(base, "rb").read()
        for _ in range(copies):
            out.write(blob)
    print("# wrote %s (%.1f GB) in %.1f s" % (big, os.path.getsize(big) / 1e9, time.time() - t0), file=sys.stderr)
host = os.path.join(ROOT, "aim_amd", "host", "host")
common = [str(n), "--algo", "wfa", "--max-score", str(ms), "--read-size", str(rs), "--reduce"] + (["--threads", threads] if threads else [])
packed = os.path.join(d, "aim_scale_%dx.aimpk" % copies)


def run(tag, inp, extra):
    out = os.path.join(d, "aim_scale_out.txt")
    best = None
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([host, inp, out] + common + extra, capture_output=True, text=True, cwd=d)
        dt = time.time() - t0
        if r.returncode != 0:
            print(json.dumps({"run": tag, "rc": r.returncode, "stderr": r.stderr[-400:], "stdout": r.stdout[-400:]}), flush=True)
            return
        best = dt if best is None else min(best, dt)
    m = re.search(r"parse\+pack ([\d.]+) ms \(line index ([\d.]+) ms\), wait ([\d.]+) ms, format\+write ([\d.]+) ms, loop ([\d.]+) ms", r.stdout)
    ph = dict(zip(("parse_pack_ms", "line_index_ms", "wait_ms", "format_write_ms", "loop_ms"), map(float, m.groups()))) if m else {}
    tm = {k: float(v) for k, v in re.findall(r"(CPU-DPU|DPU Kernel|DPU-CPU): ([\d.]+) ms", r.stdout)}
    steady = re.search(r"steady ([\d.eE+-]+) pairs/s", r.stdout)
    print(json.dumps({"run": tag, "pairs": n, "wall_s": best, "pairs_per_s_wall": n / best, "steady_pairs_per_s": float(steady.group(1)) if steady else None,
                      **ph, "device_ms": tm, "out_bytes": os.path.getsize(out), "tail": r.stdout.strip().splitlines()[-1][:300]}), flush=True)


if only in (None, "text"):
    run("text score-only", big, [])
    run("text cigar", big, ["--backtrace"])
if only in (None, "packed"):
    t0 = time.time()
    r = subprocess.run([host, big, "/dev/null"] + common + ["--pack-only", packed], capture_output=True, text=True, cwd=d)
    print("# pack-only: rc %d, %.1f s, %s" % (r.returncode, time.time() - t0, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]), file=sys.stderr)
    if r.returncode == 0:
        run("packed score-only", packed, ["--packed-input"])
        run("packed cigar", packed, ["--packed-input", "--backtrace"])
if "--keep" not in args:
    for f in (big, packed, os.path.join(d, "aim_scale_out.txt")):
        if os.path.exists(f):
            os.remove(f)
