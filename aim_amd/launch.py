#!/usr/bin/env python3
"""Counterpart of the reference launchers run-{nw,swg,wfa}-pim-{wram,mram}.py.

Same command-line flags (-i -o -l -e -n -m -x -g -a -b -r -t -d), same cost validation and the same
MAX_SCORE / READ_SIZE derivation (WFA/DPU-WRAM/run-wfa-pim-wram.py:36-68,
NW/DPU-WRAM/run-nw-pim-wram.py:31-57, SWG/DPU-WRAM/run-swg-pim-wram.py:36-62).  What the reference then
does -- `make clean; make NR_DPUS= NR_TASKLETS= FLAGS=...` (a recompile per configuration) and the
WRAM/MRAM tasklet sizing search (run-wfa-pim-wram.py:70-116) -- is replaced: the values travel to
aim_amd/host/host as run-time flags and LDS/HBM scratch is planned inside libaim_hip.so.

    python -m aim_amd.launch wfa -i pairs.seq -o out -l 100 -e 0.01 -n 20000 -b -r [-d NR_DPUS] [--gpus N]
"""
import argparse
import math
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_BIN = os.path.join(HERE, "host", "host")


def _parser(algo):
    ap = argparse.ArgumentParser(prog="aim_amd.launch " + algo)
    ap.add_argument("-i", "--input", type=str, required=True, help="Input read pairs file path")
    ap.add_argument("-o", "--output", type=str, help="Output alignment file path", default="./out")
    ap.add_argument("-l", "--read_length", required=True, type=int, help="Read Length")
    ap.add_argument("-e", "--error", type=float, required=True, help="Percentage error per read length")
    ap.add_argument("-n", "--number_reads", type=int, required=True, help="Number of read pairs to be aligned")
    ap.add_argument("-m", "--match_cost", type=int, default=0, help="Cost of characters match")
    ap.add_argument("-x", "--mismatch_cost", type=int, default=3, help="Cost of characters mismatch")
    if algo == "nw":
        ap.add_argument("-g", "--gap", type=int, default=4, help="Cost of gap deletion/insertion")
    else:
        ap.add_argument("-g", "--gap_opening", type=int, default=4, help="Cost of opening a new gap")
        ap.add_argument("-a", "--gap_extending", type=int, default=1, help="Cost of extending gap")
    ap.add_argument("-b", "--backtrace", action="store_true", help="Enable backtracing")
    if algo == "wfa":
        ap.add_argument("-r", "--reduced", action="store_true", help="Enable WFA-Adaptive")
    if algo == "genasm":
        ap.add_argument("-r", "--reduced", action="store_true", help="accepted and ignored")
    ap.add_argument("-t", "--nr_of_tasklets", type=int, help="accepted for compatibility; tasklets do not exist on MI355X")
    ap.add_argument("-d", "--nr_of_dpus", type=int, help="logical NR_DPUs of the reference partition rule (default=1)")
    ap.add_argument("--gpus", type=int, default=1, help="MI355X devices to shard over")
    ap.add_argument("--mram", action="store_true", help="SWG only: int16 cells like SWG/DPU-MRAM")
    ap.add_argument("--slots", type=int, default=0, help="batches in flight per device (host --slots; default: the host's 2, genasm 4): each slot pins its own host and device buffers")
    ap.add_argument("--dry-run", action="store_true", help="print the host command and exit")
    return ap


def parse(algo, argv):
    """Validated configuration dict (exits like the reference on bad input)."""
    args = vars(_parser(algo).parse_args(argv))
    m, x = args["match_cost"], args["mismatch_cost"]
    if algo == "nw":
        g, a = args["gap"], None
        bad = m > 0 or x <= 0 or g <= 0
    else:
        g, a = args["gap_opening"], args["gap_extending"]
        bad = m > 0 or x <= 0 or g <= 0 or a <= 0
    if bad:
        print("Wrong affine gap penalties must be  m <= 0 and g, a, x > 0\n")
        sys.exit(-1)
    if args["read_length"] <= 0:
        print("Undefined input read length")
        sys.exit(-1)
    if args["number_reads"] <= 0:
        print("Undefined number of input reads")
        sys.exit(-1)
    nr_of_wrong_bases = args["read_length"] * args["error"]
    gap_term = nr_of_wrong_bases * g if algo == "nw" else nr_of_wrong_bases * (g + a)
    max_score = math.ceil(max(nr_of_wrong_bases * x, gap_term))
    read_size = math.ceil((((args["read_length"] + nr_of_wrong_bases) + 7) / 8)) * 8
    return dict(algo=algo, input=args["input"], output=args["output"], n=args["number_reads"], match=m, mismatch=x,
                gap_o=g, gap_e=a, max_score=int(max_score), read_size=int(read_size), backtrace=args["backtrace"],
                reduce=bool(args.get("reduced")), nr_dpus=args["nr_of_dpus"] or 1, gpus=args["gpus"],
                swg_w16=args["mram"], dry_run=args["dry_run"], slots=args.get("slots") or 0)


def flag_line(cfg):
    """The -D flag list the reference launcher would have passed to make (WRAM_SEGMENT omitted)."""
    s = "-DMAX_SCORE=%d -DREAD_SIZE=%d -DMATCH=%d -DMISMATCH=%d" % (cfg["max_score"], cfg["read_size"], cfg["match"],
                                                                     cfg["mismatch"])
    if cfg["algo"] == "nw":
        s += " -DGAP_D=%d -DGAP_I=%d" % (cfg["gap_o"], cfg["gap_o"])
    else:
        s += " -DGAP_O=%d -DGAP_E=%d" % (cfg["gap_o"], cfg["gap_e"])
    if cfg["reduce"]:
        s += " -DREDUCE"
    if cfg["backtrace"]:
        s += " -DBACKTRACE"
    return s


def host_command(cfg):
    cmd = [HOST_BIN, cfg["input"], cfg["output"], str(cfg["n"]), "--algo", cfg["algo"], "--max-score",
           str(cfg["max_score"]), "--read-size", str(cfg["read_size"]), "--match", str(cfg["match"]), "--mismatch",
           str(cfg["mismatch"]), "--nr-dpus", str(cfg["nr_dpus"]), "--gpus", str(cfg["gpus"])]
    if cfg["algo"] == "nw":
        cmd += ["--gap", str(cfg["gap_o"])]
    else:
        cmd += ["--gap-o", str(cfg["gap_o"]), "--gap-e", str(cfg["gap_e"])]
    if cfg["backtrace"]:
        cmd.append("--backtrace")
    if cfg["reduce"] and cfg["algo"] == "wfa":
        cmd.append("--reduce")
    if cfg["swg_w16"]:
        cmd.append("--swg-w16")
    # batches in flight per device (host --slots). GenASM defaults to four: a pair that loses the diagonal keeps one wavefront busy long after its batch is
    # done and the next batches run under that tail (16 384 pairs of 100 kb through the CLI: 1 / 2 / 4 slots = 1.3 / 2.1 / 4.3e5 pairs/s).
    # Every slot carries its own pinned host AND device buffers (both sequence arrays + results / CIGAR): four slots are four times the pinned memory
    # per device, times --gpus -- `--slots N` lowers it on a small host.
    slots = cfg.get("slots") or (4 if cfg["algo"] == "genasm" else 0)
    if slots:
        cmd += ["--slots", str(slots)]
    return cmd


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("nw", "swg", "wfa", "genasm"):
        print("usage: python -m aim_amd.launch {nw,swg,wfa,genasm} -i IN -l LEN -e ERR -n N [options]")
        return 2
    cfg = parse(argv[0], argv[1:])
    print("run-time configuration:", flag_line(cfg))
    cmd = host_command(cfg)
    print(" ".join(cmd))
    if cfg["dry_run"]:
        return 0
    if not os.path.exists(HOST_BIN):
        print("host binary missing: run `python -m aim_amd.build`")
        return 1
    return subprocess.call(cmd)


if __name__ == "__main__":
    sys.exit(main())
