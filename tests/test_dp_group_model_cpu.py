"""CPU model of dp_group_kernel's row arithmetic (aim_amd/csrc/dp_group.hpp, round 5) against the oracle's NW / SWG (oracle/aim_oracle.c = nw.c:109-153,
swg.c:121-171 over the flat table): G lanes of K cells per pair, several pairs in one "wavefront" of 64 lanes, the in-row gap chain as a prefix minimum
D[v] = v e + min_{j < v} G[j] with G[j] = A[j] + o + e - (j + 1) e, the exclusive prefix minimum over a pair's lanes as ONE unsegmented scan over keys
(P - 1 - slot) << 16 | (value + 0x8000), cells beyond the row masked to +inf, and for plen > tlen the aliased boundary cell B(h + 1) = cell (h, W) whose
D / R value is the chain one column further: the prefix minimum over ALL of the owner lane's cells + W steps. Pure Python; pins the derivation the kernel rests on
(scores; the CIGAR walk is dp_strip's, tested on the GPU)."""
import random

import numpy as np

INF16 = 0x7FFF
BIG = 0x3FFFFFFF


def model_wave(pairs, algo, K, G, x, o, e, gi, gd, ms):
    """pairs: up to 64 // G (pattern, text) tuples sharing a wavefront. Returns their scores."""
    swg = algo == "swg"
    P = 64 // G
    assert len(pairs) <= P
    ge = e if swg else gd
    oe = o + e
    st = []
    for q in range(P):
        p, t = pairs[q] if q < len(pairs) else ("", "")
        plen, tlen = len(p), len(t)
        W = tlen + 1
        Rr = min(plen, W - 1)
        st.append(dict(p=p, t=t, plen=plen, tlen=tlen, W=W, Rr=Rr, tail=plen >= W and plen > 0 and tlen > 0,
                       M=[(o + v * e) if swg else v * gd for v in range(G * K + 1)], I=[ms] * (G * K + 1), BMprev=0, nB=None, upM_prev=((o + Rr * e) if swg else Rr * gd), last=None))
    hmax = max(s["tlen"] for s in st)
    for h in range(1, hmax + 1):
        lane_min, rows = [BIG] * 64, []
        for q, s in enumerate(st):                    # pre-carry, per lane
            if s["tail"] and h > 1:
                BM, BI, BD = s["nB"]
            else:
                BM, BI, BD = ((o + h * e, o + h * e, ms) if swg else (h * gi, 0, 0))
            A, Iv, Gv = {}, {}, {}
            tch = s["t"][h - 1] if h <= s["tlen"] else "\0"
            for v in range(1, G * K + 1):
                pch = s["p"][v - 1] if v <= s["plen"] else "\0"
                diag = s["M"][v - 1] if v > 1 else s["BMprev"]
                sub = diag + ((x if pch != tch else 0))
                ins = min(s["M"][v] + oe, s["I"][v] + e) if swg else s["M"][v] + gi
                A[v], Iv[v] = min(sub, ins), ins
                c1 = v * ge + ((e - oe) if swg else 0)
                Gv[v] = A[v] - c1 if v <= s["Rr"] else INF16
            for g in range(G):
                lane_min[q * G + g] = min(Gv[v] for v in range(g * K + 1, g * K + K + 1))
            rows.append((BM, BI, BD, A, Iv, Gv, tch))
        # ONE exclusive scan over (rank, value) keys: lanes of the pairs on the left lose every minimum; a pair's first lane sees a foreign key
        keys = [(((P - 1 - lane // G) if lane // G < P else P + 1) << 16) | (min(lane_min[lane], INF16) + 0x8000) for lane in range(64)]
        pre, run = [], BIG
        for lane in range(64):
            pre.append(run)
            run = min(run, keys[lane])
        for q, s in enumerate(st):                    # post-carry
            BM, BI, BD, A, Iv, Gv, tch = rows[q]
            rank = P - 1 - q
            newM, newI = s["M"][:], s["I"][:]
            c_owner = None
            for g in range(G):
                sk = pre[q * G + g]
                lane_pre = (sk & 0xFFFF) - 0x8000 if (sk >> 16) == rank else BIG
                c = min((min(BD, BM + o) if swg else BM), lane_pre)
                for v in range(g * K + 1, g * K + K + 1):
                    D = c + v * ge
                    newM[v] = min(A[v], D)
                    newI[v] = Iv[v]
                    c = min(c, Gv[v])
                if g == (s["Rr"] - 1) // K:
                    c_owner = c                       # the prefix minimum over ALL of the owner lane's cells (cells beyond the row: +inf)
            upM = newM[s["Rr"]] if s["Rr"] >= 1 else 0
            if s["tail"] and h <= s["tlen"]:
                pchW = s["p"][s["W"] - 1]
                cD = c_owner + s["W"] * ge            # D / R of cell (h, W): the chain one column further
                if swg:
                    cI = min(BM + oe, BI + e)
                    cM = min(s["upM_prev"] + (0 if pchW == tch else x), min(cI, cD))
                else:
                    cI = BM + gi
                    cM = min(s["upM_prev"] + (0 if pchW == tch else x), min(cI, cD))
                s["nB"] = (cM, cI, cD)
                if h == s["tlen"]:
                    s["last"] = dict(M=newM, I=newI, BM=BM, BI=BI, diag=s["upM_prev"], cD0=cD)
            elif h == s["tlen"]:
                s["last"] = dict(M=newM, I=newI, BM=BM, BI=BI)
            s["upM_prev"] = upM
            if h <= s["tlen"]:
                s["M"], s["I"], s["BMprev"] = newM, newI, BM
    out = []
    for q in range(len(pairs)):
        s = st[q]
        L = s["last"]
        plen, tlen, W = s["plen"], s["tlen"], s["W"]
        if not s["tail"]:
            out.append(L["M"][plen])
            continue
        # the last row's tail cells v = W .. plen, sequentially, with the aliased inputs (dp_strip.hpp): up = the last row's early cells, left = the previous tail cell
        tch = s["t"][tlen - 1]
        M, I, bM, bI = L["M"], L["I"], L["BM"], L["BI"]
        upM = M[W - 1]
        upD_first = L["cD0"]                         # D of cell (tlen, W) = min(upM + oe, upD + e) = the chain's next value
        lastM = 0
        for v in range(W, plen + 1):
            if v == W:
                leftM, leftI, diagM = bM, bI, L["diag"]
            else:
                leftM, leftI = M[v - W], I[v - W]
                diagM = bM if v - 1 == W else M[v - 1 - W]
            pch = s["p"][v - 1]
            if swg:
                cD = upD_first if v == W else min(upM + oe, upD + e)
                cI = min(leftM + oe, leftI + e)
                cM = min(diagM + (0 if pch == tch else x), min(cI, cD))
            else:
                cI, cD = leftM + gi, upM + gd
                cM = min(diagM + (0 if pch == tch else x), min(cI, cD))
            upM, upD, lastM = cM, cD, cM
        out.append(lastM)
    return out


def _draw_pair(rng, lo, hi):
    tl = rng.randint(lo, hi)
    t = [rng.choice("ACGT") for _ in range(tl)]
    r = rng.random()
    if r < 0.15:
        p = [rng.choice("ACGT") for _ in range(rng.randint(max(1, tl // 2), min(hi, 2 * tl)))]     # unrelated
    else:
        p = [ch if rng.random() > 0.08 else rng.choice("ACGT") for ch in t]
        for _ in range(rng.randint(0, 12)):
            if rng.random() < 0.5 and len(p) > 2: del p[rng.randrange(len(p))]
            elif len(p) < min(hi, 2 * tl): p.insert(rng.randrange(len(p) + 1), rng.choice("ACGT"))
    return "".join(p[: min(hi, 2 * tl)]), "".join(t)


def test_dp_group_model_scores_equal_the_oracle(built):
    from oracle import oracle
    rng = random.Random(23)
    K = 8                                                  # (the kernel: 32; the recurrences do not depend on it)
    for algo, G, rs, cost in (("nw", 6, 48, dict(mismatch=3, gap_i=4, gap_d=4)), ("nw", 9, 72, dict(mismatch=2, gap_i=5, gap_d=3)), ("nw", 32, 256, dict(mismatch=7, gap_i=2, gap_d=6)),
                              ("swg", 6, 48, dict(mismatch=3, gap_o=4, gap_e=1)), ("swg", 11, 88, dict(mismatch=5, gap_o=2, gap_e=3)), ("swg", 21, 168, dict(mismatch=4, gap_o=6, gap_e=2))):
        P = 64 // G
        n = 6 * P
        seqs = [_draw_pair(rng, max(4, rs // 3), rs - 8) for _ in range(n)]
        seqs[0] = (seqs[0][1] + "ACGTTGCA"[: min(8, rs - 8 - len(seqs[0][1]))], seqs[0][1])          # plen > tlen by a few
        seqs[1] = (seqs[1][1][:-1] if len(seqs[1][1]) > 4 else seqs[1][1], seqs[1][1])                  # plen == tlen - 1
        plen = np.array([len(p) for p, _ in seqs], dtype=np.int32); tlen = np.array([len(t) for _, t in seqs], dtype=np.int32)
        assert (plen > tlen + 1).any() and (plen <= tlen).any() and (plen <= 2 * tlen).all()
        pat = np.zeros((n, rs), dtype=np.uint8); txt = np.zeros((n, rs), dtype=np.uint8)
        for i, (p, t) in enumerate(seqs):
            pat[i, :len(p)] = np.frombuffer(p.encode(), dtype=np.uint8); txt[i, :len(t)] = np.frombuffer(t.encode(), dtype=np.uint8)
        ms = 2000
        op = oracle.params(algo, ms, rs, swg_cell_bytes=2 if algo == "swg" else 0, **cost)
        ref, _, _ = oracle.align_batch(op, plen, tlen, pat, txt)
        x, o, e = cost["mismatch"], cost.get("gap_o", 0), cost.get("gap_e", 0)
        gi, gd = cost.get("gap_i", 0), cost.get("gap_d", 0)
        for u in range(0, n, P):
            got = model_wave(seqs[u: u + P], algo, K, G, x, o, e, gi, gd, ms)
            for i, sc in enumerate(got):
                assert sc == int(ref["score"][u + i]) and int(ref["status"][u + i]) == 0, (algo, G, u + i, sc, int(ref["score"][u + i]), len(seqs[u + i][0]), len(seqs[u + i][1]))
