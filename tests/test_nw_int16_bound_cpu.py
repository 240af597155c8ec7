"""The bound behind dp_wave_exact_ok() for NW (aim_amd/csrc/dp_wave.hpp, round 6; NOTES R6.7): with costs >= 0 no store of the reference's flat table
(NW/DPU-WRAM/dpu/nw.c:109-153, restated in oracle/aim_oracle.c nw_pair -- row 0 over v = 0 .. plen, column 0, then every row over v = 1 .. plen with stride
tlen + 1, so that cells beyond a row alias the next rows) exceeds max(plen, tlen) * M + g, M = max(min(x, gi + gd), gi, gd), g = max(gi, gd), and no candidate
the reference casts to int16 exceeds that + max(x, g). Checked here on the recurrence itself in unbounded integers: every length relation (incl. plen >> tlen,
where a row wraps the table many times), all-mismatch and random sequences, zero and lopsided costs. The kernels' condition is (READ_SIZE + 4) M + 2 x + 2 g <
32 000; tests/test_gpu_parity.py::test_nw_long_reads_below_the_int16_bound_leave_the_literal_path runs the kernels at READ_SIZE 7 904 against the oracle."""
import random


def _nw_max(p, t, x, gi, gd):
    """The reference's fill over a flat table in Python integers: (largest stored value, largest candidate)."""
    plen, tlen = len(p), len(t)
    W = tlen + 1
    dp = [0] * (W * (tlen + 1) + plen + 2)
    hi_store = hi_cand = 0
    c = 0
    for v in range(1, plen + 1):
        c += gd
        dp[v] = c
        hi_store = max(hi_store, c)
    c = 0
    for h in range(1, tlen + 1):
        c += gi
        dp[W * h] = c
        hi_store = max(hi_store, c)
    for h in range(1, tlen + 1):
        for v in range(1, plen + 1):
            d = dp[W * h + v - 1] + gd
            i = dp[W * (h - 1) + v] + gi
            m = dp[W * (h - 1) + v - 1] + (0 if p[v - 1] == t[h - 1] else x)
            cell = min(m, i, d)
            dp[W * h + v] = cell
            hi_cand = max(hi_cand, d, i, m)
            hi_store = max(hi_store, cell)
            assert cell >= 0
    return hi_store, hi_cand


def test_no_nw_store_exceeds_read_size_times_the_largest_step():
    rng = random.Random(606)
    worst = 0.0
    for case in range(2500):
        rs = rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 40])
        plen, tlen = rng.randint(1, rs), rng.randint(1, rs)
        r = rng.random()
        if r < 0.15: tlen = 1
        elif r < 0.30: plen = rs; tlen = max(1, rs // rng.choice([2, 3, 5, 9]))
        elif r < 0.40: plen = 1
        elif r < 0.55: plen = tlen = rs
        x, gi, gd = rng.randint(0, 9), rng.randint(0, 9), rng.randint(0, 9)
        if rng.random() < 0.2: gi = gd
        kind = rng.random()
        if kind < 0.35: p, t = "A" * plen, "C" * tlen                         # every diagonal step a mismatch
        elif kind < 0.5: p, t = "A" * plen, "A" * tlen
        else: p, t = "".join(rng.choice("ACGT") for _ in range(plen)), "".join(rng.choice("ACGT") for _ in range(tlen))
        hi_store, hi_cand = _nw_max(p, t, x, gi, gd)
        g = max(gi, gd)
        M = max(min(x, gi + gd), g)
        n = max(plen, tlen)
        assert hi_store <= n * M + g, (case, plen, tlen, x, gi, gd, hi_store)
        assert hi_cand <= n * M + g + max(x, g), (case, plen, tlen, x, gi, gd, hi_cand)
        if n * M: worst = max(worst, hi_store / (n * M + g))
    assert worst > 0.9          # the bound is reached to within 10 %: it is not loose by a factor again


def test_the_kernels_condition_is_the_bound_with_a_margin():
    """dp_wave_exact_ok as the plan sees it: dp_strip_kernel up to the last READ_SIZE the condition admits, the literal path from the next one on, for three cost sets."""
    import ctypes as C
    from aim_amd import capi, engine
    lib = capi.load()
    for costs in (dict(), dict(mismatch=7, gap_i=5, gap_d=3), dict(mismatch=2, gap_i=9, gap_d=1), dict(mismatch=20, gap_i=3, gap_d=3)):
        x = costs.get("mismatch", 3); gi = costs.get("gap_i", 4); gd = costs.get("gap_d", 4)
        g = max(gi, gd); M = max(min(x, gi + gd), g)
        last = max(rs for rs in range(2568, 32000, 8) if (rs + 4) * M + 2 * x + 2 * g < 32000)
        for rs, want in ((last, b"dp_strip_kernel"), (last + 8, b"dp_wave_kernel")):
            if rs > 16384 and want == b"dp_strip_kernel":
                continue            # (beyond dp_strip's shapes the row-scan kernel takes over whatever the bound says)
            assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 10, rs, **costs))) == want, (costs, rs)
