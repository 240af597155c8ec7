#!/usr/bin/env python3
"""The host CLI at scale: wall clock and phase times of `aim_amd/host/host` on a large input file --
text and packed input, score-only and CIGAR. The input is a seeded 1 Mi-pair file replicated `copies` times (content repeats,
pair indices do not), written under /tmp (or $AIM_SCALE_DIR); one JSON line per run.

    python tools/cli_scale.py [copies=64] [--threads T] [--only text|packed] [--keep] [--shards 1,2,4,8] [--null] [--device-ids 0,0,0,0] [--check]

--shards: also run with `--out-shards K` for every K listed (K lanes, K output files); --null: output to /dev/null (host-side rate without
any file system); --check: md5 of `cat` of the shards against the single file's.
"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import engine

args = sys.argv[1:]
copies = int(args[0]) if args and args[0].isdigit() else 64
threads = args[args.index("--threads") + 1] if "--threads" in args else None
only = args[args.index("--only") + 1] if "--only" in args else None
shard_list = [int(x) for x in args[args.index("--shards") + 1].split(",")] if "--shards" in args else [1]
to_null = "--null" in args
dev_ids = args[args.index("--device-ids") + 1] if "--device-ids" in args else None
check = "--check" in args
single_md5 = {}
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


def md5_of(files):
    import hashlib
    h = hashlib.md5()
    for f in files:
        with open(f, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
    return h.hexdigest()


def run(tag, inp, extra):
    for k in shard_list:
        run_one(tag + (" shards=%d" % k if k > 1 else ""), inp, extra, k)


def run_one(tag, inp, extra, shards):
    import glob
    out = "/dev/null" if to_null and shards == 1 else os.path.join(d, "aim_scale_out.txt")
    extra = extra + (["--out-shards", str(shards)] if shards > 1 else []) + (["--device-ids", dev_ids] if dev_ids else [])
    if to_null and shards > 1:      # K lanes, every shard a symlink to /dev/null
        for k in range(shards):
            f = "%s.%03d" % (out, k)
            if os.path.lexists(f):
                os.remove(f)
            os.symlink("/dev/null", f)
    best, steadies = None, []
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([host, inp, out] + common + extra, capture_output=True, text=True, cwd=d)
        dt = time.time() - t0
        if r.returncode != 0:
            print(json.dumps({"run": tag, "rc": r.returncode, "stderr": r.stderr[-400:], "stdout": r.stdout[-400:]}), flush=True)
            return
        best = dt if best is None else min(best, dt)
        st = re.search(r"steady ([\d.eE+-]+) pairs/s", r.stdout)
        steadies.append(float(st.group(1)) if st else None)
    m = re.search(r"parse\+pack ([\d.]+) ms \(line index ([\d.]+) ms\), wait ([\d.]+) ms, format\+write ([\d.]+) ms, loop ([\d.]+) ms", r.stdout)
    ph = dict(zip(("parse_pack_ms", "line_index_ms", "wait_ms", "format_write_ms", "loop_ms"), map(float, m.groups()))) if m else {}
    tm = {k: float(v) for k, v in re.findall(r"(CPU-DPU|DPU Kernel|DPU-CPU): ([\d.]+) ms", r.stdout)}
    files = [out] if shards == 1 else sorted(glob.glob(out + ".[0-9][0-9][0-9]"))
    rec = {"run": tag, "pairs": n, "shards": shards, "wall_s": best, "pairs_per_s_wall": n / best, "steady_pairs_per_s": max(x for x in steadies if x is not None),
           "steady_runs": steadies, **ph, "device_ms": tm, "out_bytes": sum(os.path.getsize(f) for f in files), "output": "/dev/null" if to_null else "files",
           "tail": r.stdout.strip().splitlines()[-1][-420:]}
    if check and not to_null:
        key = tag.split(" shards=")[0]
        digest = md5_of(files)
        if shards == 1:
            single_md5[key] = digest
        rec["md5"] = digest
        rec["md5_equals_single_file"] = (digest == single_md5.get(key)) if key in single_md5 else None
    print(json.dumps(rec), flush=True)
    if shards > 1:
        for f in files:
            os.remove(f)


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
