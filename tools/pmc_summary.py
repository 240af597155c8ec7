#!/usr/bin/env python3
"""rocprofv3 evidence for ONE kernel of ONE command, in the format of profiles/r01/wfa_lane_pmc_summary.json:
  * a --kernel-trace --stats pass (average launch duration; the stats CSV is copied next to the summary),
  * PMC passes, each in its own run with --kernel-trace only (gpurun refuses --pmc combined with other trace domains):
    FETCH_SIZE | WRITE_SIZE | SQ instruction counts | SQ cycle / wait counters.
HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled when --fetch-x2 is given (state the access
pattern that justifies it with --note); WRITE_SIZE is exact for streaming stores.

usage (on the GPU box; the profiled program comes straight after `--`, no env/bash hop):
  python3 tools/pmc_summary.py --out profiles/r02/wfa_lane_pmc_summary.json --kernel wfa_lane_kernel --pairs 4194304 \
      --alg-bytes 905968812 --fetch-x2 --io compact -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
"""
import argparse, collections, csv, glob, json, os, shutil, subprocess, sys

PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT"],
    # lane use of the vector instructions: thread-cycles over instruction-cycles x 64 (rocprof's VALUUtilization)
    ["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--pairs", type=int, required=True, help="pairs one launch processes")
    ap.add_argument("--alg-bytes", type=float, default=None, help="algorithmic bytes per launch (SURVEY 8d definition)")
    ap.add_argument("--fetch-x2", action="store_true", help="apply the guide's gfx950 correction (16 B/lane coalesced streaming reads)")
    ap.add_argument("--io", default=None, help="wire layout tag recorded in the summary (bench.py matches it)")
    ap.add_argument("--note", default="")
    ap.add_argument("--plan", default=None, help="plan line to record (default: the one the profiled program prints; pass it when the program "
                                                 "prints another kernel's plan, e.g. bench.py's headline plan while the e2e leg's packed kernel is profiled)")
    ap.add_argument("--skip-first", type=int, default=0, help="ignore the first N launches of the kernel (warm-up)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    assert cmd, "no command"
    os.environ.setdefault("TMPDIR", "/tmp")
    work = os.path.join("gpurun_out", "pmc_%d" % os.getpid())
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    summary = {"command": "rocprofv3 --kernel-trace [--stats | --pmc <set>] -- " + " ".join(cmd), "kernel_match": a.kernel,
               "pairs_per_launch": a.pairs, "raw": {}}
    if a.io:
        summary["io"] = a.io

    # ---- pass 0: kernel trace + stats
    d = os.path.join(work, "stats")
    r0 = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd,
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=1800, text=True)
    for line in (r0.stdout or "").splitlines():   # the profiled program may state its own algorithmic bytes / plan (tools/bench_configs.py)
        if line.startswith("{"):
            try:
                j = json.loads(line)
            except Exception:
                continue
            if a.alg_bytes is None and "algorithmic_bytes" in j:
                a.alg_bytes = float(j["algorithmic_bytes"])
            if "plan" in j and a.plan is None:
                summary["plan"] = j["plan"]
            if "config" in j and isinstance(j["config"], dict) and "plan" in j["config"] and a.plan is None:
                summary["plan"] = j["config"]["plan"]
    if a.plan is not None:
        summary["plan"] = a.plan
    durs, disp = [], None
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if a.kernel in r["Kernel_Name"]:
                durs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
                disp = disp or {"kernel": r["Kernel_Name"], "grid_threads": int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                                "workgroup": int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                                "vgpr": int(r.get("VGPR_Count", 0) or 0), "accum_vgpr": int(r.get("Accum_VGPR_Count", 0) or 0),
                                "sgpr": int(r.get("SGPR_Count", 0) or 0), "lds_bytes": int(r.get("LDS_Block_Size", 0) or 0),
                                "scratch_bytes": int(r.get("Scratch_Size", 0) or 0)}
    durs = [x[1] for x in sorted(durs)][a.skip_first:]
    if durs:
        summary["kernel"] = disp["kernel"]
        disp["lds_note"] = "rocprofv3's LDS_Block_Size is the STATIC allocation; these kernels use dynamic LDS only -- its size is the `lds=` field of the plan line"
        summary["dispatch"] = disp
        disp["vgpr_note"] = ("rocprofv3's VGPR_Count is the dispatch record's rounded ARCH count; the kernel's allocation is `code_object` below "
                             "(.vgpr_count / .agpr_count / .private_segment_fixed_size of the library's metadata notes, tools/codeobj_regs.py)")
        try:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import codeobj_regs
            summary["code_object"] = codeobj_regs.lookup(codeobj_regs.kernel_regs(), disp["kernel"])
        except Exception as e:   # (no llvm tools: the summary says so instead of guessing)
            summary["code_object"] = {"error": repr(e)}
        summary["kernel_trace"] = {"launches": len(durs), "avg_us": sum(durs) / len(durs) / 1e3, "min_us": min(durs) / 1e3, "max_us": max(durs) / 1e3}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, os.path.splitext(a.out)[0].replace("_pmc_summary", "") + "_kernel_stats.csv")
    shutil.rmtree(d, ignore_errors=True)

    # ---- PMC passes
    for cs in PASSES:
        d = os.path.join(work, "pmc")
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + cs + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1800)
        per = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch id -> value
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if a.kernel in r["Kernel_Name"]:
                    per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for c, byd in per.items():
            vals = [byd[k] for k in sorted(byd)][a.skip_first:]
            if vals:
                summary["raw"][c] = {"per_launch_mean": sum(vals) / len(vals), "launches": len(vals)}
        shutil.rmtree(d, ignore_errors=True)
    shutil.rmtree(work, ignore_errors=True)

    raw = summary["raw"]
    if "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
        fetch = raw["FETCH_SIZE"]["per_launch_mean"] * 1024.0
        write = raw["WRITE_SIZE"]["per_launch_mean"] * 1024.0
        summary["fetch_bytes_raw"] = fetch
        summary["fetch_bytes_corrected_x2"] = fetch * 2 if a.fetch_x2 else None
        summary["write_bytes"] = write
        summary["hbm_traffic_bytes_per_launch"] = (fetch * 2 if a.fetch_x2 else fetch) + write
        summary["correction"] = ("MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a "
                                 "16-B-per-lane coalesced streaming read (applied: %s); WRITE_SIZE is exact for streaming stores. %s"
                                 % ("yes" if a.fetch_x2 else "no -- access width uncalibrated, raw value kept", a.note))
    if a.alg_bytes:
        summary["algorithmic_bytes_per_launch"] = a.alg_bytes
        if summary.get("hbm_traffic_bytes_per_launch") and summary["hbm_traffic_bytes_per_launch"] < a.alg_bytes:
            summary["traffic_note"] = ("the counters' traffic (%.1f MB) is BELOW the bytes this kernel must read and write (%.1f MB algorithmic): the x2 rule for FETCH_SIZE is "
                                       "calibrated on 16-B-per-lane coalesced streaming reads and under-counts this kernel's loads (narrower or strided accesses are counted at a "
                                       "different granularity) -- the figure is a lower bound, not a measurement of re-use" % (summary["hbm_traffic_bytes_per_launch"] / 1e6, a.alg_bytes / 1e6))
        if "kernel_trace" in summary:
            summary["algorithmic_GBps"] = a.alg_bytes / (summary["kernel_trace"]["avg_us"] * 1e-6) / 1e9
            summary["roofline_frac_of_8TBps"] = summary["algorithmic_GBps"] / 8000.0
    if "SQ_WAVE_CYCLES" in raw:
        wc = raw["SQ_WAVE_CYCLES"]["per_launch_mean"]
        summary["derived"] = {k.lower() + "_over_wave_cycles": raw[k]["per_launch_mean"] / wc
                              for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in raw}
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"):
            if k in raw:
                summary["derived"][k.lower() + "_per_pair"] = raw[k]["per_launch_mean"] / a.pairs
    if "SQ_THREAD_CYCLES_VALU" in raw and raw.get("SQ_ACTIVE_INST_VALU", {}).get("per_launch_mean"):
        summary.setdefault("derived", {})["valu_lane_use"] = raw["SQ_THREAD_CYCLES_VALU"]["per_launch_mean"] / (raw["SQ_ACTIVE_INST_VALU"]["per_launch_mean"] * 64.0)
    json.dump(summary, open(a.out, "w"), indent=1)
    print(json.dumps({k: summary.get(k) for k in ("kernel", "kernel_trace", "hbm_traffic_bytes_per_launch", "algorithmic_GBps")}))


if __name__ == "__main__":
    main()
