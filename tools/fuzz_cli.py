#!/usr/bin/env python3
"""Randomised differential run of the drop-in CLI path: `python -m aim_amd.launch` (launcher arithmetic -> C host program
-> HIP kernels -> output writer) against oracle/oracle_cli (the CPU restatement of the reference launcher + host + kernels)
on random input files with the reference parser's quirks (no final newline, unpaired last line, n that is not a cap,
logical NR_DPUS, non-ACGT bytes, over-length reads). Output files must be byte-identical and exit codes equal.
The oracle is the checker only.

    python tools/fuzz_cli.py [--seconds 120] [--seed 1]"""
import argparse, json, os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aim_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--focus", choices=["all", "dplong"], default="all", help="'dplong': long-read NW/SWG with CIGAR (the family of the one unexplained difference)")
a = ap.parse_args()
rng = random.Random(a.seed)
cli = os.path.join(ROOT, "oracle", "oracle_cli")
t0, cases, skipped = time.time(), 0, 0
with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None) as td:
    while time.time() - t0 < a.seconds:
        algo = rng.choice(["nw", "nw", "swg"]) if a.focus == "dplong" else rng.choice(["wfa", "wfa", "nw", "swg"])
        l = rng.choice([300, 700, 700]) if a.focus == "dplong" else rng.choice([20, 64, 100, 100, 150, 300, 700])
        e = rng.choice([0.0, 0.01, 0.02, 0.05, 0.1])
        pairs = rng.choice([9, 40, 200, 1000, 3000]) if l <= 150 else rng.choice([9, 40, 200])
        cost = {}
        if rng.random() < 0.4:
            cost = dict(x=rng.randint(1, 6), g=rng.randint(1, 6))
            if algo != "nw": cost["a"] = rng.randint(1, 3)
        ms, rs = engine.launcher_sizes(algo, l, e, **({"mismatch": cost.get("x", 3), "gap_o": cost.get("g", 4), "gap_e": cost.get("a", 1), "gap": cost.get("g", 4)}))
        req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, pairs, l, e, max(rs, l + 64))
        data = bytearray(engine.pairs_to_text(req, pat, txt))
        quirk = rng.choice(["plain", "plain", "no_final_newline", "unpaired_tail", "non_acgt", "overlong"])
        if quirk == "no_final_newline": data = data[:-1]
        elif quirk == "unpaired_tail": data += b">ACGTACGT\n"
        elif quirk == "non_acgt":
            for _ in range(5):
                i = rng.randrange(len(data))
                if data[i] not in b"\n><": data[i] = ord(rng.choice("Nnx*"))
        elif quirk == "overlong":
            lines = bytes(data).split(b"\n"); k = rng.randrange(max(1, len(lines) - 1))
            lines[k] = lines[k] + b"A" * (rs + 8); data = bytearray(b"\n".join(lines))
        inp = os.path.join(td, "in.seq"); open(inp, "wb").write(bytes(data))
        d = rng.choice([1, 1, 2, 3, 4, 8])
        n = rng.choice([pairs, pairs // 2 + 1, pairs * 2, d, d + 1, 1])
        flags = []
        if a.focus == "dplong" or rng.random() < 0.6: flags.append("-b")
        if algo == "wfa" and rng.random() < 0.6: flags.append("-r")
        cflags = []
        for k, v in cost.items(): cflags += ["-" + k, str(v)]
        oh, oo = os.path.join(td, "h.out"), os.path.join(td, "o.out")
        for f in (oh, oo):
            if os.path.exists(f): os.remove(f)
        common = ["-i", inp, "-l", str(l), "-e", str(e), "-n", str(n), "-d", str(d)] + flags + cflags
        rh = subprocess.run([sys.executable, "-m", "aim_amd.launch", algo, "-o", oh] + common, capture_output=True, text=True, cwd=td,
                            env=dict(os.environ, PYTHONPATH=ROOT))
        ro = subprocess.run([cli, algo, "-o", oo] + common, capture_output=True, text=True, cwd=td)
        case = dict(algo=algo, l=l, e=e, pairs=pairs, n=n, d=d, quirk=quirk, flags=flags, cost=cost, rc_host=rh.returncode, rc_oracle=ro.returncode)
        bh = open(oh, "rb").read() if os.path.exists(oh) else None
        bo = open(oo, "rb").read() if os.path.exists(oo) else None
        same_rc = (rh.returncode == 0) == (ro.returncode == 0)
        ok = same_rc and (rh.returncode != 0 or bh == bo)
        cases += 1
        if not ok:
            print(json.dumps(dict(case, ok=False)), flush=True)
            print("HOST stdout tail:", rh.stdout[-300:], "stderr:", rh.stderr[-300:], flush=True)
            print("ORACLE stdout tail:", ro.stdout[-300:], "stderr:", ro.stderr[-300:], flush=True)
            if bh is not None and bo is not None: print("output sizes", len(bh), len(bo), flush=True)
            keep = os.path.join(ROOT, "gpurun_out", "fuzz_cli_fail"); os.makedirs(keep, exist_ok=True)
            import shutil
            for f in (inp, oh, oo):
                if os.path.exists(f): shutil.copy(f, keep)
            if bh is not None and bo is not None:
                lh, lo = bh.split(b"\n"), bo.split(b"\n")
                diff = [i for i in range(min(len(lh), len(lo))) if lh[i] != lo[i]]
                print("differing lines:", len(diff), "first:", diff[:5], flush=True)
                for i in diff[:3]: print("  line", i, "host", lh[i][:120], "| oracle", lo[i][:120], flush=True)
            sys.exit(1)
print(json.dumps({"cases": cases, "seconds": round(time.time() - t0, 1), "all_ok": True}), flush=True)
