"""CPU model of the two recurrences genasm_wave.hpp runs, checked against the textbook GenASM-DC recurrence (the one oracle/genasm_oracle.c
restates): (1) the BANDED, complemented 32-bit words of the fast path -- exact on the bits the hit test and the traceback read, for any
window shape; (2) the full-width path's scan over columns -- the (k, A, B) composition, the per-step shifts of a Hillis-Steele scan built
from DPP row_shr 1/2/4/8 + row_bcast 15/31, the initial column folded in at the end. Pure Python integers; no GPU, no library: this pins
the DERIVATION, the GPU tests pin the kernel."""
import random

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def textbook(p, t, levels):
    """R[a][d] of GenASM-DC (Algorithm 1) for window pattern p, text t; column n = the initial ~0 << d."""
    m, n = len(p), len(t)
    R = [[0] * levels for _ in range(n + 1)]
    for d in range(levels):
        R[n][d] = (M64 << d) & M64
    for a in range(n - 1, -1, -1):
        pm = M64
        for j in range(m):
            if p[m - 1 - j] == t[a]:
                pm &= ~(1 << j)
        R[a][0] = ((R[a + 1][0] << 1) | pm) & M64
        for d in range(1, levels):
            R[a][d] = (((R[a + 1][d] << 1) | pm) & (R[a + 1][d - 1] << 1) & R[a + 1][d - 1] & (R[a][d - 1] << 1)) & M64
    return R


def eq_mask(p, ch):
    m = len(p)
    v = 0
    for q in range(m):
        if p[m - 1 - q] == ch:
            v |= 1 << q
    return v


def lowmask(k):
    return 0 if k <= 0 else (M32 if k >= 32 else (1 << k) - 1)


def band_eq(eq, col, m):           # gl_band_eq
    s = m - 16 - col
    if s >= 0:
        return (eq >> s) & M32
    up = -s
    return M32 if up >= 32 else ((eq << up) | ((1 << up) - 1)) & M32


def banded(p, t):
    """c[a][d], d = 0..15: bit b <-> full bit b + (m-16-a), complemented; column n's word is the 16 + n - m + d lowest bits."""
    m, n = len(p), len(t)
    nm = 16 + n - m
    C = {n: [lowmask(nm + d) for d in range(16)]}
    for a in range(n - 1, -1, -1):
        e = band_eq(eq_mask(p, t[a]), a, m)
        row = []
        for d in range(16):
            c = C[a + 1][d] & e
            if d:
                cp, cc = C[a + 1][d - 1], row[d - 1]
                c |= cp | ((cc << 1) & M32) | (cp >> 1)
            row.append(c & M32)
        C[a] = row
    return C


def random_window(rng):
    m = rng.choice([64, 64, 64, 50, 33, 17, 5, 1])
    n = rng.choice([64, 64, 64, 40, 29, 5, 64])
    p = [rng.choice("ACGT") for _ in range(m)]
    if rng.random() < 0.7:       # related: the pattern with edits
        t = []
        e = rng.choice([0.05, 0.1, 0.2, 0.3])
        for ch in p:
            r = rng.random()
            if r < e / 3:
                continue
            if r < 2 * e / 3:
                t.append(rng.choice("ACGT"))
            t.append(rng.choice("ACGT") if r < e else ch)
        t = (t + [rng.choice("ACGT") for _ in range(64)])[:n]
    else:
        t = [rng.choice("ACGT") for _ in range(n)]
    return p, t


def test_banded_words_are_exact_where_they_are_read():
    rng = random.Random(11)
    for _ in range(120):
        p, t = random_window(rng)
        m, n = len(p), len(t)
        R, C = textbook(p, t, 16), banded(p, t)
        for a in range(n + 1):
            for d in range(16):
                for b in range(d, 31 - d):          # at level d the bits d .. 30 - d are exact; everything the walk reads lies inside
                    q = b + m - 16 - a
                    if q >= m:
                        continue                    # positions before the pattern's start: never read
                    want = 1 if q < 0 else 1 - ((R[a][d] >> q) & 1)   # past the pattern's end: "aligns"
                    assert (C[a][d] >> b) & 1 == want, (m, n, a, d, b)


def _scan_steps():
    """(source lane or None, shift) per lane for row_shr:1/2/4/8, row_bcast:15 (rows 1, 3), row_bcast:31 (rows 2, 3)."""
    out = []
    for kind in (1, 2, 4, 8, "b15", "b31"):
        src, k2 = [None] * 64, [0] * 64
        for j in range(64):
            i, r = j % 16, j // 16
            if kind in (1, 2, 4, 8):
                if i >= kind:
                    src[j], k2[j] = j - kind, kind
            elif kind == "b15":
                if r in (1, 3):
                    src[j], k2[j] = 16 * r - 1, i + 1
            elif r in (2, 3):
                src[j], k2[j] = 31, (j % 32) + 1
        out.append((src, k2))
    return out


def full_width_scan(p, t, levels):
    """C[d][lane], lane j <-> column n-1-j, as ga_dc64_scan computes it."""
    m, n = len(p), len(t)
    EQ = [eq_mask(p, t[n - 1 - j]) if n - 1 - j >= 0 else 0 for j in range(64)]
    steps = _scan_steps()
    A, ak = list(EQ), []
    for src, k2 in steps:
        ak.append(list(A))
        A = [(((A[src[j]] << k2[j]) & M64) & A[j]) if src[j] is not None else A[j] for j in range(64)]
    out, prev = [], None
    for d in range(levels):
        if d == 0:
            B = [e & 1 for e in EQ]
        else:
            B = []
            for j in range(64):
                cp = prev[j - 1] if j else (1 << (d - 1)) - 1
                B.append(((cp << 1) | cp | (prev[j] << 1) | 1) & M64)
        for si, (src, k2) in enumerate(steps):
            B = [B[j] | ((((B[src[j]] << k2[j]) & M64) & ak[si][j]) if src[j] is not None else 0) for j in range(64)]
        init = (1 << d) - 1
        prev = [((0 if j == 63 else (init << (j + 1)) & M64) & A[j]) | B[j] for j in range(64)]
        out.append(prev)
    return out


def test_full_width_scan_equals_the_recurrence():
    rng = random.Random(12)
    for _ in range(25):
        p, t = random_window(rng)
        m, n = len(p), len(t)
        R, C = textbook(p, t, 64), full_width_scan(p, t, 64)
        mask = (1 << m) - 1
        for d in range(64):
            for a in range(n):
                assert C[d][n - 1 - a] & mask == (~R[a][d]) & mask, (m, n, d, a)


def walk_textbook(R, p, t, last, levels):
    """GenASM-TB of one window over the full vectors, the oracle's rule order (oracle/genasm_oracle.c); None: no alignment within `levels`."""
    m, n = len(p), len(t)
    d = next((k for k in range(levels) if not (R[0][k] >> (m - 1)) & 1), None)
    if d is None:
        return None
    ok = lambda r, b: b >= m or not (r >> (m - 1 - b)) & 1
    a = b = 0
    ops = []
    while True:
        if b == m or (not last and (a >= 40 or b >= 40)):
            break
        if a == n:
            ops.append("D"); b += 1; d -= 1; continue
        if p[b] == t[a] and ok(R[a + 1][d], b + 1):
            ops.append("M"); a += 1; b += 1; continue
        if d > 0 and ok(R[a + 1][d - 1], b + 1):
            ops.append("X"); a += 1; b += 1; d -= 1; continue
        if d > 0 and ok(R[a][d - 1], b + 1):
            ops.append("D"); b += 1; d -= 1; continue
        if d > 0 and ok(R[a + 1][d - 1], b):
            ops.append("I"); a += 1; d -= 1; continue
        raise AssertionError("no rule applies")
    return "".join(ops), a, b


def walk_banded(C, p, t, last):
    """walk_any of genasm_wave.hpp over the banded words: the run of matches is bit kb = 15 + ca - cb of (c_{a+1}[d] & eq_a) in consecutive
    columns; the edit after it is decided by bits kb, kb - 1, kb + 1 of level d - 1 at the one cell where the run stopped."""
    m, n = len(p), len(t)
    d = next((k for k in range(16) if (C[0][k] >> 15) & 1), None)
    if d is None:
        return None
    ca = cb = 0
    ops = []
    while True:
        if ca == n:
            k = (m - cb) if last else (0 if ca >= 40 else max(min(m, 40) - cb, 0))
            ops += ["D"] * k; cb += k; d -= k
            break
        kb = 15 + ca - cb
        lim = min(n - ca, m - cb)
        if not last:
            lim = min(lim, 40 - max(ca, cb))
        run = 0
        while run < lim and (C[ca + run + 1][d] & band_eq(eq_mask(p, t[ca + run]), ca + run, m)) >> kb & 1:
            run += 1
        ops += ["M"] * run; ca += run; cb += run
        if cb == m or (not last and (ca >= 40 or cb >= 40)):
            break
        if ca == n:
            continue
        assert d > 0
        s1, s0 = C[ca + 1][d - 1], C[ca][d - 1]
        op = "X" if (s1 >> kb) & 1 else "D" if (s0 >> ((kb - 1) & 31)) & 1 else "I" if (s1 >> ((kb + 1) & 31)) & 1 else None
        assert op is not None
        ops.append(op); ca += op != "D"; cb += op != "I"; d -= 1
    return "".join(ops), ca, cb


def test_banded_walk_equals_the_textbook_walk():
    rng = random.Random(13)
    seen_last = seen_irregular = 0
    for _ in range(400):
        p, t = random_window(rng)
        last = rng.random() < 0.4
        R, C = textbook(p, t, 16), banded(p, t)
        want = walk_textbook(R, p, t, last, 16)
        got = walk_banded(C, p, t, last)
        assert want == got, (len(p), len(t), last, want, got)
        if want is not None:
            seen_last += last; seen_irregular += (len(p) != 64 or len(t) != 64)
    assert seen_last > 20 and seen_irregular > 20
