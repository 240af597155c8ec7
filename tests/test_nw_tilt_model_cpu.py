"""CPU model of nw_reg_kernel's TILTED recurrence (dp_reg.hpp, round 4) against the oracle's NW (oracle/aim_oracle.c = nw.c:67-153):
T(h, v) = R(h, v) - GAP_I * h - GAP_D * v. The move from the row above costs nothing, the chain along the row is a running minimum,
the diagonal carries mismatch - GAP_I - GAP_D; score and traceback (two direction bits per cell, made at fill time from the tilted
candidates) must come out as the oracle's for pairs without aliased cells (plen <= tlen). Pure Python; pins the derivation, the GPU
tests pin the kernel."""
import random

import numpy as np


def tilted_nw(p, t, x, gi, gd):
    plen, tlen = len(p), len(t)
    INF = 16000
    prev = [0] * (plen + 1)                      # row 0: R(0, v) = v * GAP_D -> 0
    bits = [[0] * (plen + 1) for _ in range(tlen + 1)]
    for h in range(1, tlen + 1):
        cur = [0] * (plen + 1)
        left = INF                               # left of column 0
        for v in range(0, plen + 1):
            diag = prev[v - 1] if v else INF
            sub = diag + (x if (v and p[v - 1] != t[h - 1]) else 0) - gi - gd
            ins = prev[v]
            a = min(sub, ins)
            m = min(a, left)
            bits[h][v] = (1 if a < left else 0) | (2 if sub < ins else 0)   # bit 0: "not D" (the chain lost strictly), bit 1: "not I"
            cur[v] = m
            left = m
        prev = cur
    score = prev[plen] + gi * tlen + gd * plen
    ops, h, v = [], tlen, plen                   # nw_traceback's order: D first, then I, else X / M
    while h > 0 and v > 0:
        c = bits[h][v]
        if not c & 1:
            ops.append("D"); v -= 1
        elif not c & 2:
            ops.append("I"); h -= 1
        else:
            ops.append("X" if p[v - 1] != t[h - 1] else "M"); h -= 1; v -= 1
    ops += ["I"] * h + ["D"] * v
    return score, "".join(reversed(ops))


def test_tilted_recurrence_gives_the_oracles_score_and_cigar(built):
    from oracle import oracle
    rng = random.Random(5)
    rs = 64
    for x, gi, gd in ((3, 4, 4), (2, 5, 5), (4, 2, 7), (9, 1, 3)):
        n = 60
        plen = np.zeros(n, dtype=np.int32); tlen = np.zeros(n, dtype=np.int32)
        pat = np.zeros((n, rs), dtype=np.uint8); txt = np.zeros((n, rs), dtype=np.uint8)
        seqs = []
        for i in range(n):
            tl = rng.randint(1, 40)
            pl = rng.randint(1, tl)              # plen <= tlen: no aliased cells (quirk N1)
            t = [rng.choice("ACGT") for _ in range(tl)]
            p = [ch if rng.random() > 0.2 else rng.choice("ACGT") for ch in t[:pl]]
            if rng.random() < 0.5 and pl > 3:
                del p[rng.randrange(len(p))]
            pl = len(p)
            plen[i], tlen[i] = pl, tl
            pat[i, :pl] = np.frombuffer("".join(p).encode(), dtype=np.uint8); txt[i, :tl] = np.frombuffer("".join(t).encode(), dtype=np.uint8)
            seqs.append((p, t))
        op = oracle.params("nw", 0, rs, mismatch=x, gap=gi, backtrace=True)
        op.gap_i, op.gap_d = gi, gd
        res, ops, worst = oracle.align_batch(op, plen, tlen, pat, txt)
        assert worst == 0
        for i, (p, t) in enumerate(seqs):
            score, cig = tilted_nw(p, t, x, gi, gd)
            assert score == int(res["score"][i]), (i, x, gi, gd)
            b, e = int(res["begin_offset"][i]), int(res["end_offset"][i])
            assert cig == bytes(ops[i, b:e]).decode(), (i, x, gi, gd)
