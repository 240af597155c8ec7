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


def write_packed(out, seed, num_pairs, length, error, batch):
    """Packed batch file (aim_amd/host/host.c, pkfile_hdr_t): 64-byte header, then per batch {n, ascii = 0, n_raw, READ_SIZE} +
    aim_request8_t[n] + packed patterns + packed texts + raw side list (indices, ASCII patterns, ASCII texts)."""
    import math
    import numpy as np
    read_size = int(math.ceil((length + length * error + 7) / 8)) * 8          # run-*-pim-*.py: READ_SIZE
    batch = max(1, min(batch, max(num_pairs, 1)))
    hdr = np.zeros(64, dtype=np.uint8)
    hdr[:8] = np.frombuffer(b"AIMPK\0\0\1", dtype=np.uint8)
    hdr[8:24] = np.array([1, read_size, 8, batch], dtype="<u4").view(np.uint8)
    hdr[24:32] = np.array([num_pairs], dtype="<u8").view(np.uint8)
    out.write(hdr.tobytes())
    for first in range(0, num_pairs, batch):
        n = min(batch, num_pairs - first)
        req, pat, txt = engine.gen_pairs(seed, first, n, length, error, read_size)
        pp, pt, raw, rawp, rawt = engine.pack_batch(req, pat, txt)
        out.write(np.array([n, 0, len(raw), read_size], dtype="<u4").tobytes())
        out.write(engine.to_request8(req).tobytes())
        for arr in (pp, pt, raw.astype("<u4"), rawp, rawt):
            out.write(np.ascontiguousarray(arr).tobytes())


def main(argv=None):
    ap = argparse.ArgumentParser(prog="aim_amd.gen_dataset", description=__doc__.split("\n\n")[0])
    ap.add_argument("-n", "--num-pairs", type=int, required=True)
    ap.add_argument("-l", "--length", type=int, required=True, help="pattern length")
    ap.add_argument("-e", "--error", type=float, required=True, help="error rate, e.g. 0.01")
    ap.add_argument("-s", "--seed", type=int, default=42)
    ap.add_argument("-o", "--output", required=True, help="output file ('-' for stdout)")
    ap.add_argument("--chunk", type=int, default=1 << 16, help="pairs generated per chunk (memory bound)")
    ap.add_argument("--packed", action="store_true",
                    help="write a packed batch file (2 bits per base + raw side list, the format `host --packed-input` reads and "
                         "`host --pack-only` writes) instead of text; READ_SIZE by the launchers' rule for (-l, -e)")
    ap.add_argument("--batch", type=int, default=1 << 20, help="--packed: pairs per batch of the file")
    a = ap.parse_args(argv)
    if a.num_pairs < 0 or a.length <= 0 or not (0.0 <= a.error < 1.0):
        ap.error("need num-pairs >= 0, length > 0, 0 <= error < 1")
    # a text can outgrow the pattern by at most the number of edits; rows are 8-byte multiples like READ_SIZE
    edits = int(-(-a.length * a.error // 1))
    row = (a.length + edits + 1 + 7) // 8 * 8
    out = sys.stdout.buffer if a.output == "-" else open(a.output, "wb")
    if a.packed:
        try:
            write_packed(out, a.seed, a.num_pairs, a.length, a.error, a.batch)
        finally:
            if out is not sys.stdout.buffer:
                out.close()
        return 0
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
