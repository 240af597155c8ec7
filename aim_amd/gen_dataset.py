"""Dataset generator CLI: the counterpart of smarco/WFA's `generate_dataset` that produced AIM's sample files
(Datasets/README.md:18-26). Writes AIM's input format -- per pair a '>'pattern line and a '<'text line -- from the same
seeded generator bench.py and the tests use (`aim_gen_pairs`: pattern = `length` uniform A/C/G/T, text = pattern after
ceil(length*error) sequential uniform edits; splitmix64 keyed on (seed, pair index), so any slice of a data set can be
regenerated independently). No GPU needed.

    python -m aim_amd.gen_dataset -n 40000 -l 100 -e 0.01 -o sample-l100-e1-40K [-s 42]
"""
import argparse
import sys

from . import engine


def main(argv=None):
    ap = argparse.ArgumentParser(prog="aim_amd.gen_dataset", description=__doc__.split("\n\n")[0])
    ap.add_argument("-n", "--num-pairs", type=int, required=True)
    ap.add_argument("-l", "--length", type=int, required=True, help="pattern length")
    ap.add_argument("-e", "--error", type=float, required=True, help="error rate, e.g. 0.01")
    ap.add_argument("-s", "--seed", type=int, default=42)
    ap.add_argument("-o", "--output", required=True, help="output file ('-' for stdout)")
    ap.add_argument("--chunk", type=int, default=1 << 16, help="pairs generated per chunk (memory bound)")
    a = ap.parse_args(argv)
    if a.num_pairs < 0 or a.length <= 0 or not (0.0 <= a.error < 1.0):
        ap.error("need num-pairs >= 0, length > 0, 0 <= error < 1")
    # a text can outgrow the pattern by at most the number of edits; rows are 8-byte multiples like READ_SIZE
    edits = int(-(-a.length * a.error // 1))
    row = (a.length + edits + 1 + 7) // 8 * 8
    out = sys.stdout.buffer if a.output == "-" else open(a.output, "wb")
    try:
        for first in range(0, a.num_pairs, a.chunk):
            n = min(a.chunk, a.num_pairs - first)
            req, pat, txt = engine.gen_pairs(a.seed, first, n, a.length, a.error, row)
            out.write(engine.pairs_to_text(req, pat, txt))
    finally:
        if out is not sys.stdout.buffer:
            out.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
