"""CPU model of the FOUR DIRECTION BITS per cell that dp_strip_kernel / dp_group_kernel decide at fill time (NW: two), and of the walk over them
(aim_amd/csrc/dp_strip.hpp: dp_traceback_swg_bits), against the oracle's swg_traceback / nw_traceback (oracle/aim_oracle.c = swg.c:45-119, nw.c:67-107 over
the reference's flat table of stride W = tlen + 1, whose rows alias for plen > tlen). Pure Python, one pair at a time; pins the derivation BASELINE config 4's
CIGAR rests on:
  * bit "M != D" = A < D, bit "M != I" = A < I (A = min(diagonal + cost, I)), bit "I extended" = I_up + e < M_up + o + e, bit "next D extended" =
    pre(v) < G(v) kept at the cell on the LEFT of the one it belongs to (the prefix-minimum form of the in-row chain, dp_strip.hpp);
  * boundary cells (column 0): analytic, or -- plen > tlen -- the previous row's tail cell (h - 1, W), whose bits go to a byte per row; the last row's tail
    cells v = W .. plen live in row tlen + 1 of the canonical table;
  * the walk: canonical position (R, C) of flat index W h + v, layers M / I / D (NW: no layers), 'M' / 'X' from the characters the cell was computed with."""
import random

import numpy as np

INF = 1 << 28


def fill_bits(p, t, swg, x, o, e, gi, gd, ms, match=0):
    plen, tlen = len(p), len(t)
    W = tlen + 1
    Rr = min(plen, W - 1)
    tail = plen >= W
    oe = o + e
    ge = e if swg else gd
    M = [(o + v * e) if swg else v * gd for v in range(Rr + 1)]
    I = [ms] * (Rr + 1)
    M[0] = 0
    bits = {}                                        # (row, column) -> (nD, nI, xD_next, xI); column 0: (nD, nI, xD_own, xI, d1ext)
    BMprev, upM_prev = 0, M[Rr]
    nB = None
    for h in range(1, tlen + 1):
        if tail and h > 1:
            BM, BI, BD = nB
        else:
            BM, BI, BD = ((o + h * e, o + h * e, ms) if swg else (h * gi, 0, 0))
            if swg:
                bits[(h, 0)] = (0, 0, 0, 0, 0 if (o + h * e) + o <= ms else 1)      # row-init boundary: only "D of column 1 extended" is ever asked
        tch = t[h - 1]
        pre = min(BD, BM + o) if swg else BM         # G[0]
        newM, newI = [BM] + [0] * Rr, [BI] + [0] * Rr
        lastD = None
        for v in range(1, Rr + 1):
            diag = M[v - 1] if v > 1 else BMprev
            sub = diag + (match if p[v - 1] == tch else x) if swg else diag + (0 if p[v - 1] == tch else x)
            if swg:
                insn, inse = M[v] + oe, I[v] + e
                ins, xI = min(insn, inse), int(inse < insn)
            else:
                ins, xI = M[v] + gi, 0
            A = min(sub, ins)
            c1 = v * ge + ((e - oe) if swg else 0)
            G = A - c1
            D = pre + v * ge
            xDn = int(pre < G) if swg else 0         # "the NEXT cell's D was extended", kept here
            bits[(h, v)] = (int(A < D), int(A < ins), xDn, xI)
            newM[v], newI[v] = min(A, D), ins
            pre = min(pre, G)
            lastD = D
        if tail:                                     # cell (h, W) = B(h + 1): D / R is the chain one column further
            cD = pre + W * ge
            if swg:
                cI = min(BM + oe, BI + e)
                cM = min(upM_prev + (match if p[W - 1] == tch else x), min(cI, cD))
            else:
                cI = BM + gi
                cM = min(upM_prev + (0 if p[W - 1] == tch else x), min(cI, cD))
            nB = (cM, cI, cD)
            if h < tlen:
                xD = bits[(h, Rr)][2]                # = upD + e < upM + o + e
                bits[(h + 1, 0)] = (int(cM != cD), int(cM != cI), xD, int(BI + e < BM + oe) if swg else 0, (0 if (not swg or cM + o <= cD) else 1))
            last_tail = dict(BM=BM, BI=BI, diag=upM_prev, upM=newM[Rr], upD=lastD)
        upM_prev = newM[Rr]
        BMprev = BM
        M, I = newM, newI
    score = M[plen] if not tail else None
    if tail:                                         # the last row's tail cells v = W .. plen: row tlen + 1, columns C = v - W
        h = tlen
        tch = t[h - 1]
        bM, bI = last_tail["BM"], last_tail["BI"]
        upM, upD = last_tail["upM"], last_tail["upD"]
        pend_xD = {}
        for v in range(W, plen + 1):
            if v == W:
                leftM, leftI, diagM = bM, bI, last_tail["diag"]
            else:
                leftM, leftI = M[v - W], I[v - W]
                diagM = bM if v - 1 == W else M[v - 1 - W]
            if swg:
                cD, cI = min(upM + oe, upD + e), min(leftM + oe, leftI + e)
                cM = min(diagM + (match if p[v - 1] == tch else x), min(cI, cD))
                xD, xI = int(upD + e < upM + oe), int(leftI + e < leftM + oe)
            else:
                cI, cD = leftM + gi, upM + gd
                cM = min(diagM + (0 if p[v - 1] == tch else x), min(cI, cD))
                xD = xI = 0
            C = v - W
            if C == 0:
                bits[(h + 1, 0)] = (int(cM != cD), int(cM != cI), xD, xI, (0 if (not swg or cM + o <= cD) else 1))
            else:
                bits[(h + 1, C)] = (int(cM != cD), int(cM != cI), 0, xI)
                if C >= 2:
                    pend_xD[C - 1] = xD              # kept at the cell on its left
            upM, upD, score = cM, cD, cM
        for C, xD in pend_xD.items():
            b = bits[(h + 1, C)]
            bits[(h + 1, C)] = (b[0], b[1], xD, b[3])
    return score, bits


def walk(p, t, swg, bits):
    plen, tlen = len(p), len(t)
    W = tlen + 1
    h, v = tlen, plen
    f = W * h + v
    R, C = f // W, f % W
    layer, ops = 0, []
    while h > 0 and v > 0:
        if C >= 1:
            nD, nI, _, xI = bits[(R, C)][:4]
            xD = 0
            if layer == 2:
                xD = bits[(R, 0)][4] if C == 1 else bits[(R, C - 1)][2]
        else:
            nD, nI, xD, xI = bits[(R, 0)][:4]
        if not swg and not nD: layer = 2
        elif not swg and not nI: layer = 1
        if layer == 2:
            ops.append("D")
            if not swg or not xD: layer = 0
            v -= 1
            if C > 0: C -= 1
            else: R -= 1; C = W - 1
        elif layer == 1:
            ops.append("I")
            if not swg or not xI: layer = 0
            h -= 1; R -= 1
        elif not nD: layer = 2
        elif not nI: layer = 1
        else:
            as_tail = C == 0 or R > tlen
            pc = p[W + C - 1] if as_tail else p[C - 1]
            tc = t[R - 2] if as_tail else t[R - 1]
            ops.append("X" if pc != tc else "M")
            h -= 1; v -= 1; R -= 1
            if C > 0: C -= 1
            else: R -= 1; C = W - 1
    ops += ["I"] * h + ["D"] * v
    return "".join(reversed(ops))


def _draw(rng, lo, hi):
    tl = rng.randint(lo, hi)
    t = [rng.choice("ACGT") for _ in range(tl)]
    if rng.random() < 0.15:
        p = [rng.choice("ACGT") for _ in range(rng.randint(max(1, tl // 2), min(hi, 2 * tl)))]
    else:
        p = [ch if rng.random() > 0.08 else rng.choice("ACGT") for ch in t]
        for _ in range(rng.randint(0, 14)):
            if rng.random() < 0.5 and len(p) > 2: del p[rng.randrange(len(p))]
            elif len(p) < min(hi, 2 * tl): p.insert(rng.randrange(len(p) + 1), rng.choice("ACGT"))
    return "".join(p[: min(hi, 2 * tl)]), "".join(t)


def test_direction_bits_and_their_walk_equal_the_oracle(built):
    from oracle import oracle
    rng = random.Random(41)
    rs = 96
    for algo, cost, ms in (("swg", dict(mismatch=3, gap_o=4, gap_e=1), 500), ("swg", dict(mismatch=5, gap_o=2, gap_e=3), 500), ("swg", dict(mismatch=4, gap_o=6, gap_e=2), 60),
                           ("nw", dict(mismatch=3, gap_i=4, gap_d=4), 0), ("nw", dict(mismatch=2, gap_i=5, gap_d=3), 0), ("nw", dict(mismatch=7, gap_i=2, gap_d=6), 0)):
        n = 150
        seqs = [_draw(rng, 8, rs - 8) for _ in range(n)]
        plen = np.array([len(p) for p, _ in seqs], dtype=np.int32); tlen = np.array([len(t) for _, t in seqs], dtype=np.int32)
        assert (plen > tlen + 1).any() and (plen == tlen).any() and (plen < tlen).any() and (plen <= 2 * tlen).all()
        pat = np.zeros((n, rs), dtype=np.uint8); txt = np.zeros((n, rs), dtype=np.uint8)
        for i, (p, t) in enumerate(seqs):
            pat[i, :len(p)] = np.frombuffer(p.encode(), dtype=np.uint8); txt[i, :len(t)] = np.frombuffer(t.encode(), dtype=np.uint8)
        swg = algo == "swg"
        op = oracle.params(algo, ms, rs, backtrace=True, swg_cell_bytes=2 if swg else 0, **cost)
        ref, rops, _ = oracle.align_batch(op, plen, tlen, pat, txt)
        x, o, e = cost["mismatch"], cost.get("gap_o", 0), cost.get("gap_e", 0)
        for i, (p, t) in enumerate(seqs):
            assert int(ref["status"][i]) == 0
            score, bits = fill_bits(p, t, swg, x, o, e, cost.get("gap_i", 0), cost.get("gap_d", 0), ms)
            assert score == int(ref["score"][i]), (algo, i, score, int(ref["score"][i]), len(p), len(t))
            want = rops[i, int(ref["begin_offset"][i]): int(ref["end_offset"][i])].tobytes().decode()
            got = walk(p, t, swg, bits)
            assert got == want, (algo, cost, i, len(p), len(t), got, want)
