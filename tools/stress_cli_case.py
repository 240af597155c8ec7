#!/usr/bin/env python3
"""Diagnostic: hunt a non-reproducible CLI difference tools/fuzz_cli.py saw once (NW, l=700, CIGAR, 80 reads over 3 logical
DPUs; equal output sizes, different content; not reproduced in 6 800 later cases). Long-read NW/SWG with CIGAR, random file
quirks / pair counts / logical DPUs like the fuzzer; each input goes through the host program TWICE and through
oracle_cli, so nondeterminism shows even without knowing the trigger. Keeps the files of the first failure."""
import os, random, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import engine
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
td = os.path.join(ROOT, "gpurun_out", "stress"); os.makedirs(td, exist_ok=True)
cli = os.path.join(ROOT, "oracle", "oracle_cli")
bad_oracle = bad_self = 0
t0 = time.time()
for it in range(iters):
    algo = rng.choice(["nw", "nw", "swg"]); l = rng.choice([300, 700, 700]); e = rng.choice([0.02, 0.05, 0.1])
    pairs = rng.choice([9, 40, 200]); d = rng.choice([1, 2, 3, 4, 8]); n = rng.choice([pairs, pairs // 2 + 1, pairs * 2, 80])
    if n <= d: n = d + 1
    ms, rs = engine.launcher_sizes(algo, l, e)
    req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, pairs, l, e, rs)
    data = bytearray(engine.pairs_to_text(req, pat, txt))
    quirk = rng.choice(["plain", "no_final_newline", "unpaired_tail", "non_acgt"])
    if quirk == "no_final_newline": data = data[:-1]
    elif quirk == "unpaired_tail": data += b">ACGTACGT\n"
    elif quirk == "non_acgt":
        for _ in range(5):
            i = rng.randrange(len(data))
            if data[i] not in b"\n><": data[i] = ord(rng.choice("Nnx*"))
    inp = os.path.join(td, "in.seq"); open(inp, "wb").write(bytes(data))
    common = ["-i", inp, "-l", str(l), "-e", str(e), "-n", str(n), "-d", str(d), "-b"]
    outs = []
    for tag in ("h1", "h2"):
        o = os.path.join(td, tag + ".out")
        r = subprocess.run([sys.executable, "-m", "aim_amd.launch", algo, "-o", o] + common, capture_output=True, text=True, cwd=td, env=dict(os.environ, PYTHONPATH=ROOT))
        outs.append((r.returncode != 0, open(o, "rb").read() if r.returncode == 0 else b""))
    oo = os.path.join(td, "o.out")
    ro = subprocess.run([cli, algo, "-o", oo] + common, capture_output=True, text=True)
    bo = (ro.returncode != 0, open(oo, "rb").read() if ro.returncode == 0 else b"")   # the reference aborts (exit 1) on some SWG pairs
    f_or = outs[0] != bo or outs[1] != bo
    f_self = outs[0] != outs[1]
    outs = [o[1] for o in outs]; bo = bo[1]
    if (f_or or f_self) and bad_oracle + bad_self == 0:
        keep = os.path.join(ROOT, "gpurun_out", "stress_fail"); os.makedirs(keep, exist_ok=True)
        for f in ("in.seq", "h1.out", "h2.out", "o.out"): shutil.copy(os.path.join(td, f), keep)
        print("FAIL at iteration", it, dict(algo=algo, l=l, e=e, pairs=pairs, n=n, d=d, quirk=quirk), flush=True)
        for name, b in (("h1", outs[0]), ("h2", outs[1])):
            lh, lo = b.split(b"\n"), bo.split(b"\n")
            diff = [i for i in range(min(len(lh), len(lo))) if lh[i] != lo[i]]
            print("  ", name, "vs oracle: differing lines", len(diff), diff[:6], flush=True)
            for i in diff[:2]: print("     line", i, "host", lh[i][:150], "\n          oracle", lo[i][:150], flush=True)
    bad_oracle += f_or; bad_self += f_self
print("iterations %d in %.0f s: differs from oracle %d, host run 1 != host run 2 %d" % (iters, time.time() - t0, bad_oracle, bad_self), flush=True)
