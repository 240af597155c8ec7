"""The bound behind dp_wave_exact_ok() for SWG (aim_amd/csrc/dp_wave.hpp): no value the reference's flat table stores, and no candidate it casts to its cell type
(SWG/DPU-WRAM/dpu/swg.c:121-171, restated in oracle/aim_oracle.c SWG_IMPL -- three layers, MAX_SCORE as +infinity in the boundary's I / D layers, stride tlen + 1 so
that cells beyond a row alias the next rows), leaves [match * n, 3 o + (2 n + 4) e + 2 x + MAX_SCORE], n = max(plen, tlen). Checked on the recurrence itself in
unbounded integers (the kernels admit a configuration when both ends, taken at n = READ_SIZE, lie inside +-32 000): every length relation incl. plen >> tlen,
all-mismatch / identical / random sequences, match <= 0, zero gap costs."""
import random


def _swg_range(p, t, match, x, o, e, maxs):
    plen, tlen = len(p), len(t)
    W = tlen + 1
    n = W * (tlen + 1) + plen + 2
    M, I, D = [0] * n, [0] * n, [0] * n
    lo = hi = 0
    def see(*vals):
        nonlocal lo, hi
        lo = min(lo, *vals); hi = max(hi, *vals)
    D[0] = I[0] = maxs
    for v in range(1, plen + 1):
        D[v] = M[v] = o + v * e
        I[v] = maxs
    for h in range(1, tlen + 1):
        D[W * h] = maxs
        I[W * h] = M[W * h] = o + h * e
    see(maxs, o + max(plen, tlen) * e)
    for h in range(1, tlen + 1):
        for v in range(1, plen + 1):
            at = W * h + v
            dn, de = M[at - 1] + o + e, D[at - 1] + e
            D[at] = min(dn, de)
            inw, ie = M[at - W] + o + e, I[at - W] + e
            I[at] = min(inw, ie)
            mm = M[at - W - 1] + (match if p[v - 1] == t[h - 1] else x)
            M[at] = min(mm, I[at], D[at])
            see(dn, de, inw, ie, mm)
    return lo, hi


def test_no_swg_value_leaves_the_range_the_plan_checks():
    rng = random.Random(707)
    tight = 0.0
    for case in range(2000):
        rs = rng.choice([1, 2, 3, 5, 8, 13, 21, 30])
        plen, tlen = rng.randint(1, rs), rng.randint(1, rs)
        r = rng.random()
        if r < 0.15: tlen = 1
        elif r < 0.30: plen = rs; tlen = max(1, rs // rng.choice([2, 3, 5, 9]))
        elif r < 0.40: plen = 1
        elif r < 0.55: plen = tlen = rs
        match, x, o, e = rng.choice([0, 0, 0, -1, -2]), rng.randint(0, 9), rng.randint(0, 9), rng.randint(0, 5)
        maxs = rng.choice([0, 5, 25, 126, 200, 1000])
        kind = rng.random()
        if kind < 0.35: p, t = "A" * plen, "C" * tlen
        elif kind < 0.5: p, t = "A" * plen, "A" * tlen
        else: p, t = "".join(rng.choice("ACGT") for _ in range(plen)), "".join(rng.choice("ACGT") for _ in range(tlen))
        lo, hi = _swg_range(p, t, match, x, o, e, maxs)
        n = max(plen, tlen)
        bound_hi = 3 * o + (2 * n + 4) * e + 2 * x + maxs
        assert hi <= bound_hi, (case, plen, tlen, match, x, o, e, maxs, hi, bound_hi)
        assert lo >= match * n, (case, plen, tlen, match, lo)
        if bound_hi: tight = max(tight, hi / bound_hi)
    assert tight > 0.6
