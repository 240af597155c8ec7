"""CPU model of swg_reg_kernel's arithmetic (dp_reg.hpp, round 5) against the oracle's SWG (oracle/aim_oracle.c = swg.c:45-171):
int8 cells as value * 256 in 16-bit fields (the hardware's wrap IS the reference's int8 wrap), the D chain in its short form
D[v] = min(A[v-1] + o + e, D[v-1] + e), the sign of the OR of every stored M as the "something may have wrapped" test, plen > tlen pairs
with cell (h, W) as the next row's boundary cell and the LAST row's tail cells computed once, and the traceback over four direction bits
per cell decided at fill time (flat indices beyond W resolved like the kernel's walk). Pure Python; pins the derivation:
  * whenever the model does NOT flag a pair, score and CIGAR equal the oracle's;
  * whenever the oracle's own arithmetic wraps (detected by re-running it with wide cells), the model flags the pair."""
import random

import numpy as np

SC = 256


def w16(x):
    x &= 0xFFFF
    return x - 0x10000 if x & 0x8000 else x


def swg_reg_model(p, t, x, o, e, ms, tails=8):
    plen, tlen = len(p), len(t)
    W = tlen + 1
    if plen < 1 or tlen < 1 or plen > tlen + 1 + tails:
        return None
    lw = plen > tlen
    pe = W if lw else plen
    OE, E, X = (o + e) * SC, e * SC, x * SC
    acc = 0
    M = [0] * (pe + 1)
    I = [0] * (pe + 1)
    val = o * SC
    for v in range(1, pe + 1):                       # row 0
        val = w16(val + E)
        M[v], I[v] = val, w16(ms * SC)
        acc |= M[v] & 0x8000
    if lw:
        M[pe] = I[pe] = w16(OE)                      # flat cell (0, W) is B(1): the row initialisation's {o + e}
    Mb, Db, Mb_old = w16(OE), w16(ms * SC), 0
    bits = [[0] * (pe + 1) for _ in range(tlen + 1)]
    for h in range(1, tlen + 1):
        acc |= Mb & 0x8000
        Dprev, Aprev, diag_prev = Db, w16(Mb + OE), Mb_old
        newM, newI = M[:], I[:]
        for v in range(1, pe + 1):
            mm = w16(diag_prev + (X if p[v - 1] != t[h - 1] else 0))
            insn, inse = w16(M[v] + OE), w16(I[v] + E)
            ins = min(insn, inse)
            A = min(mm, ins)
            de = w16(Dprev + E)
            d = min(Aprev, de)
            m = min(A, d)
            bits[h][v] = (1 if A < d else 0) | (2 if mm < ins else 0) | (4 if d < Aprev else 0) | (8 if inse < insn else 0)
            acc |= m & 0x8000
            diag_prev = M[v]
            newM[v], newI[v] = m, ins
            Dprev, Aprev = d, w16(A + OE)
            lastM, lastD = m, d
        M, I = newM, newI
        Mb_old = Mb
        if lw:
            Mb, Db = lastM, lastD                    # cell (h, W) is B(h + 1)
        else:
            Mb = w16(Mb + E)
    score16 = M[pe]
    tailbits = {}
    if lw and plen > pe:                             # the last row's tail cells v = W + c
        upM, upD = Mb, Db
        for c in range(1, plen - pe + 1):
            mU, iU = M[c], I[c]
            dg = Mb_old if c == 1 else M[c - 1]
            dn, de, in_, ie = w16(upM + OE), w16(upD + E), w16(mU + OE), w16(iU + E)
            cD, cI = min(dn, de), min(in_, ie)
            mm = w16(dg + (X if p[pe + c - 1] != t[tlen - 1] else 0))
            cM = min(mm, min(cI, cD))
            tailbits[c] = (1 if min(mm, cI) < cD else 0) | (2 if mm < cI else 0) | (4 if de < dn else 0) | (8 if ie < in_ else 0)
            acc |= cM & 0x8000
            upM, upD, score16 = cM, cD, cM
    if acc:
        return "flagged"
    ops, h, v, layer = [], tlen, plen, 0             # swg_traceback over the bits; flat indices beyond W resolved as in the kernel's walk
    while h > 0 and v > 0:
        pi, ti = v - 1, h - 1
        if v > W and h == tlen:
            b = tailbits[v - W]
        else:
            hh, vv = (h + 1, v - W) if v > W else (h, v)
            if v > W:
                pi, ti = vv - 1, hh - 1
            b = bits[hh][vv]
        if layer == 2:
            ops.append("D"); layer = 0 if not b & 4 else 2; v -= 1
        elif layer == 1:
            ops.append("I"); layer = 0 if not b & 8 else 1; h -= 1
        elif not b & 1:
            layer = 2
        elif not b & 2:
            layer = 1
        else:
            ops.append("X" if p[pi] != t[ti] else "M"); h -= 1; v -= 1
    ops += ["I"] * h + ["D"] * v
    return score16 >> 8, "".join(reversed(ops))


def test_swg_reg_model_equals_the_oracle_or_flags_the_pair(built):
    from oracle import oracle
    rng = random.Random(11)
    rs = 72
    flagged = exact = 0
    for x, o, e, ms in ((3, 4, 1, 5), (3, 4, 1, 25), (2, 5, 1, 60), (7, 3, 2, 20), (4, 6, 2, 100), (5, 1, 3, 126)):
        n = 80
        plen = np.zeros(n, dtype=np.int32); tlen = np.zeros(n, dtype=np.int32)
        pat = np.zeros((n, rs), dtype=np.uint8); txt = np.zeros((n, rs), dtype=np.uint8)
        seqs = []
        for i in range(n):
            tl = rng.randint(12, 60)                 # (tail cells c <= 8 < W, as in the kernel: its rows are at least READ_SIZE - 38 columns long)
            t = [rng.choice("ACGT") for _ in range(tl)]
            if rng.random() < 0.25:
                p = [rng.choice("ACGT") for _ in range(rng.randint(1, min(64, tl + 8)))]       # unrelated: cells climb, int8 wraps
            else:
                p = [ch if rng.random() > 0.1 else rng.choice("ACGT") for ch in t]
                for _ in range(rng.randint(0, 3)):
                    if rng.random() < 0.5 and len(p) > 2: del p[rng.randrange(len(p))]
                    elif len(p) < min(64, tl + 8): p.insert(rng.randrange(len(p) + 1), rng.choice("ACGT"))
            p = p[: tl + 1 + 8]
            plen[i], tlen[i] = len(p), tl
            pat[i, :len(p)] = np.frombuffer("".join(p).encode(), dtype=np.uint8); txt[i, :tl] = np.frombuffer("".join(t).encode(), dtype=np.uint8)
            seqs.append((p, t))
        assert (plen > tlen + 1).any() and (plen == tlen + 1).any() and (plen <= tlen).any()
        op8 = oracle.params("swg", ms, rs, mismatch=x, gap_o=o, gap_e=e, backtrace=True, swg_cell_bytes=1)
        op16 = oracle.params("swg", ms, rs, mismatch=x, gap_o=o, gap_e=e, backtrace=True, swg_cell_bytes=2)
        r8, ops8, _ = oracle.align_batch(op8, plen, tlen, pat, txt)
        r16, ops16, _ = oracle.align_batch(op16, plen, tlen, pat, txt)
        for i, (p, t) in enumerate(seqs):
            got = swg_reg_model(p, t, x, o, e, ms)
            same_wide = int(r8["score"][i]) == int(r16["score"][i]) and int(r8["status"][i]) == 0 and int(r16["status"][i]) == 0
            if got == "flagged":
                flagged += 1
                continue
            exact += 1
            assert int(r8["status"][i]) == 0, (i, x, o, e, ms)
            score, cig = got
            assert score == int(r8["score"][i]), (i, x, o, e, ms, len(p), len(t))
            b, en = int(r8["begin_offset"][i]), int(r8["end_offset"][i])
            assert cig == bytes(ops8[i, b:en]).decode(), (i, x, o, e, ms, len(p), len(t))
            assert same_wide                         # an unflagged pair did not wrap: int16 cells give the same score
    assert flagged > 20 and exact > 200, (flagged, exact)
