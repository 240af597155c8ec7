#!/usr/bin/env python3
"""Generate tests/golden/launcher_goldens.json by RUNNING the reference's own
Python launchers (run-*-pim-*.py) in this container.

The launchers echo the full `make NR_DPUS=.. NR_TASKLETS=.. FLAGS="..."` line
before invoking make (e.g. WFA/DPU-WRAM/run-wfa-pim-wram.py:128-134); that line
is the golden for the sizing counterpart in aim_amd/launch.py.  The subsequent
`make`/`./build/host` calls fail harmlessly in the scratch cwd (no UPMEM SDK).

Only runs where /root/reference exists (not on the GPU box); the JSON it
writes is the committed fixture.
"""
import itertools
import json
import os
import re
import subprocess
import sys
import tempfile

REF = "/root/reference"
SCRIPTS = {
    ("wfa", "wram"): "WFA/DPU-WRAM/run-wfa-pim-wram.py",
    ("wfa", "mram"): "WFA/DPU-MRAM/run-wfa-pim-mram.py",
    ("nw", "wram"): "NW/DPU-WRAM/run-nw-pim-wram.py",
    ("nw", "mram"): "NW/DPU-MRAM/run-nw-pim-mram.py",
    ("swg", "wram"): "SWG/DPU-WRAM/run-swg-pim-wram.py",
    ("swg", "mram"): "SWG/DPU-MRAM/run-swg-pim-mram.py",
}

LE = [(100, 0.01), (100, 0.02), (100, 0.05), (100, 0.07), (100, 0.10), (150, 0.03), (250, 0.05),
      (500, 0.01), (1000, 0.05), (1000, 0.10), (5000, 0.02), (10000, 0.01)]
COSTS = [None, (0, 4, 6, 2), (0, 2, 3, 1), (-1, 5, 4, 2)]  # (m, x, g, a); None = defaults


def run_one(algo, variant, l, e, costs, backtrace, reduce_):
    args = [sys.executable, os.path.join(REF, SCRIPTS[(algo, variant)]), "-i", "in", "-o", "out",
            "-l", str(l), "-e", repr(e), "-n", "100000", "-d", "4"]
    if costs is not None:
        m, x, g, a = costs
        args += ["-m", str(m), "-x", str(x), "-g", str(g)]
        if algo != "nw":
            args += ["-a", str(a)]
    if backtrace:
        args.append("-b")
    if reduce_ and algo == "wfa":
        args.append("-r")
    with tempfile.TemporaryDirectory() as cwd:
        out = subprocess.run(args, cwd=cwd, capture_output=True, text=True).stdout
    line = next((ln for ln in out.splitlines() if ln.startswith("make NR_DPUS")), None)
    rec = {"algo": algo, "variant": variant, "l": l, "e": e, "costs": costs,
           "backtrace": backtrace, "reduce": bool(reduce_ and algo == "wfa"), "make_line": line,
           "stdout_head": out.splitlines()[:1]}
    if line:
        for key in ("MAX_SCORE", "READ_SIZE", "MATCH", "MISMATCH", "GAP_O", "GAP_E", "GAP_I", "GAP_D"):
            mm = re.search(r"-D%s=(-?\d+)" % key, line)
            if mm:
                rec[key] = int(mm.group(1))
        rec["REDUCE"] = "-DREDUCE" in line
        rec["BACKTRACE"] = "-DBACKTRACE" in line
    return rec


def main():
    recs = []
    for (algo, variant) in SCRIPTS:
        for (l, e), costs in itertools.product(LE, COSTS):
            if variant == "mram" and costs is not None:
                continue  # sizing flags are identical across variants; keep the fixture small
            recs.append(run_one(algo, variant, l, e, costs, True, True))
        recs.append(run_one(algo, variant, 100, 0.01, None, False, False))
        recs.append(run_one(algo, variant, 100, 0.01, (1, 3, 4, 1), True, False))   # m > 0 -> rejected
        recs.append(run_one(algo, variant, 100, 0.01, (0, 0, 4, 1), True, False))   # x <= 0 -> rejected
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "launcher_goldens.json")
    with open(dst, "w") as f:
        json.dump(recs, f, indent=0)
    print("wrote", len(recs), "records to", dst)


if __name__ == "__main__":
    main()
