"""GenASM (BASELINE config 5).  PARITY UNPINNED: the reference tree holds only an un-pinned, empty submodule
(/root/reference/.gitmodules:1-3), so there is nothing of AIM's to compare with.  What is tested: (CPU) the oracle's
restatement of the published algorithm against the defining properties of an alignment and against the exact edit
distance on small inputs; (GPU) the HIP kernel against that oracle, bit for bit, up to config 5's read length."""
import os
import numpy as np
import pytest


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def _check_alignment(req, pat, txt, res, ops):
    """The ops consume exactly the pattern (M/X/D) and the text (M/X/I), 'M' only over equal bytes, 'X' only over different
    ones, and score = number of non-M operations."""
    for i in range(len(req)):
        pl, tl = int(req["pattern_len"][i]), int(req["text_len"][i])
        assert res["begin_offset"][i] == 0 and res["status"][i] == 0
        o = ops[i, : res["end_offset"][i]]
        is_m, is_x, is_i, is_d = (o == ord("M")), (o == ord("X")), (o == ord("I")), (o == ord("D"))
        assert (is_m | is_x | is_i | is_d).all()
        adv_p = (is_m | is_x | is_d).astype(np.int64)
        adv_t = (is_m | is_x | is_i).astype(np.int64)
        assert adv_p.sum() == pl and adv_t.sum() == tl, i
        pi, ti = np.cumsum(adv_p) - adv_p, np.cumsum(adv_t) - adv_t
        diag = is_m | is_x
        assert ((pat[i, pi[diag]] == txt[i, ti[diag]]) == is_m[diag]).all(), i
        assert int((~is_m).sum()) == int(res["score"][i]), i


@pytest.mark.parametrize("l,err,n", [(1, 0.0, 4), (30, 0.1, 300), (64, 0.05, 300), (100, 0.1, 300), (300, 0.25, 60), (1000, 0.1, 40)])
def test_genasm_oracle_properties_and_upper_bound(built, l, err, n):
    from aim_amd import engine
    from oracle import oracle
    rs = ((int(l * (1 + err)) + 8 + 7) // 8) * 8
    req, pat, txt = engine.gen_pairs(7 + l, 0, n, l, err, rs)
    for i in range(0, n, 11):
        pat[i, i % l] = ord("N")                       # any byte is a character; equal bytes match
    res, ops, worst = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), req["pattern_len"], req["text_len"], pat, txt, nthreads=4)
    assert worst == 0
    _check_alignment(req, pat, txt, res, ops)
    sres, _, _ = oracle.align_batch(oracle.params("genasm", 0, rs), req["pattern_len"], req["text_len"], pat, txt, nthreads=2)
    assert np.array_equal(sres["score"], res["score"])
    if l <= 300:
        exact = 0
        for i in range(min(n, 80)):
            ed = _edit_distance(pat[i, : req["pattern_len"][i]].tobytes(), txt[i, : req["text_len"][i]].tobytes())
            assert res["score"][i] >= ed
            exact += int(res["score"][i] == ed)
        assert exact >= 0.8 * min(n, 80)              # the windowed heuristic is exact on most pairs


def test_genasm_oracle_identical_and_disjoint_sequences(built):
    from oracle import oracle
    rs = 208
    pat = np.zeros((3, rs), dtype=np.uint8); txt = np.zeros((3, rs), dtype=np.uint8)
    pat[0, :200] = ord("A"); txt[0, :200] = ord("A")                       # identical
    pat[1, :200] = ord("A"); txt[1, :200] = ord("C")                       # no character in common: the [spec] fallback window
    pat[2, :150] = ord("G"); txt[2, :60] = ord("G")                        # text much shorter
    plen = np.array([200, 200, 150], dtype=np.int32); tlen = np.array([200, 200, 60], dtype=np.int32)
    res, ops, worst = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), plen, tlen, pat, txt)
    assert worst == 0 and list(res["score"]) == [0, 200, 90]
    req = np.zeros(3, dtype=[("pattern_len", "<i4"), ("text_len", "<i4")]); req["pattern_len"] = plen; req["text_len"] = tlen
    _check_alignment(req, pat, txt, res, ops)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def gpu(built):
    from aim_amd import capi
    import ctypes as C
    lib = capi.load()
    n = C.c_int()
    assert lib.aim_device_count(C.byref(n)) == 0 and n.value >= 1, "no HIP device visible"
    return lib


def _hip_vs_oracle(l, err, n, seed, backtrace, dirty=True, **mk):
    from aim_amd import engine
    from oracle import oracle
    rs = ((int(l * (1 + err)) + 8 + 7) // 8) * 8
    req, pat, txt = engine.gen_pairs(seed, 0, n, l, err, rs)
    if dirty:
        for i in range(0, n, 5):
            pat[i, (7 * i) % l] = ord("N")
    params = engine.make_params("genasm", 0, rs, backtrace=backtrace, **mk)
    res, ops = engine.align(params, req, pat, txt)
    ores, oops, worst = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=backtrace), req["pattern_len"], req["text_len"],
                                           pat, txt, nthreads=16)
    assert worst == 0
    assert np.array_equal(res["score"], ores["score"]) and np.array_equal(res["idx"], req["idx"])
    if not mk.get("res8"):
        for f in ("max_operations", "begin_offset", "end_offset", "status"):
            assert np.array_equal(res[f], ores[f]), f
    if backtrace:
        for i in range(n):
            e = int(res["end_offset"][i])
            assert np.array_equal(ops[i, :e], oops[i, :e]), i
        _check_alignment(req, pat, txt, res, ops)
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("backtrace", [True, False])
@pytest.mark.parametrize("l,err,n", [(1, 0.0, 70), (30, 0.1, 500), (64, 0.05, 500), (100, 0.1, 2000), (300, 0.25, 300), (1000, 0.1, 300),
                                     (5000, 0.02, 64), (10000, 0.1, 40)])
def test_genasm_hip_matches_oracle(gpu, l, err, n, backtrace):
    _hip_vs_oracle(l, err, n, 100 + l, backtrace)
    if not backtrace:
        _hip_vs_oracle(l, err, min(n, 100), 200 + l, False, res8=True)


@pytest.mark.gpu
def test_genasm_cfg5_l100000_e10(gpu):
    """BASELINE config 5's shape: 100 kb reads at 10 % error (~2 600 dependent windows per pair), with CIGAR."""
    res = _hip_vs_oracle(100000, 0.10, 24, 5, True, dirty=False)
    assert res["score"].min() > 5000 and res["score"].max() < 12000


@pytest.mark.gpu
def test_genasm_through_the_host_cli(gpu, tmp_path):
    """`python -m aim_amd.launch genasm` -> C host (packed input, device-side CIGAR runs) prints the oracle's alignment."""
    import subprocess, sys
    from conftest import ROOT
    from aim_amd import engine
    from oracle import oracle
    l, err, n = 3000, 0.1, 60
    rs = int(np.ceil((l + l * err + 7) / 8)) * 8
    req, pat, txt = engine.gen_pairs(77, 0, n, l, err, rs)
    inp = tmp_path / "in.seq"
    inp.write_bytes(engine.pairs_to_text(req, pat, txt))
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, "-m", "aim_amd.launch", "genasm", "-i", str(inp), "-o", str(out), "-l", str(l), "-e", str(err),
                        "-n", str(n), "-b"], capture_output=True, text=True, cwd=tmp_path, env=dict(__import__("os").environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout + r.stderr
    ores, oops, _ = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), req["pattern_len"], req["text_len"], pat, txt, nthreads=8)
    ores["idx"] = req["idx"]
    assert out.read_bytes() == oracle.format_output(ores, oops, True)


@pytest.mark.gpu
def test_genasm_windows_that_need_more_than_15_edits(gpu):
    """The kernel's fast path computes 16 error levels per window and falls back to all 64; both sides of that switch and the
    [spec] no-alignment window are exercised: e = 35-60 % error rates, unrelated sequences, single-letter sequences."""
    from aim_amd import engine
    from oracle import oracle
    for l, err, n, seed in ((200, 0.35, 400, 1), (500, 0.6, 200, 2), (64, 0.5, 500, 3)):
        rs = ((int(l * (1 + err)) + 8 + 7) // 8) * 8
        req, pat, txt = engine.gen_pairs(seed, 0, n, l, err, rs)
        rng = np.random.RandomState(seed)
        for i in range(0, n, 3):                                   # unrelated text: random bases over the whole length
            tl = int(req["text_len"][i])
            txt[i, :tl] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=tl)
        if n > 10:
            pat[1, : req["pattern_len"][1]] = ord("A"); txt[1, : req["text_len"][1]] = ord("C")   # nothing matches: the [spec] fallback window
        params = engine.make_params("genasm", 0, rs, backtrace=True)
        res, ops = engine.align(params, req, pat, txt)
        ores, oops, worst = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), req["pattern_len"], req["text_len"], pat, txt, nthreads=16)
        assert worst == 0
        for f in ("score", "end_offset", "status"):
            assert np.array_equal(res[f], ores[f]), f
        for i in range(n):
            e = int(res["end_offset"][i])
            assert np.array_equal(ops[i, :e], oops[i, :e]), i
        _check_alignment(req, pat, txt, res, ops)
        assert (res["score"] > 16 * (req["pattern_len"] // 40 + 1) // 2).any()   # some windows are far beyond 15 edits


@pytest.mark.gpu
@pytest.mark.parametrize("l,err,n", [(30, 0.1, 300), (64, 0.05, 300), (100, 0.1, 1000), (150, 0.3, 400), (300, 0.25, 300), (1000, 0.1, 300), (500, 0.6, 100),
                                     (5000, 0.02, 64), (3000, 0.15, 64)])
def test_genasm_window_shapes_match_oracle(gpu, monkeypatch, l, err, n):
    """Short last windows, windows with m != n, windows beyond 15 edits, with and without ops -- same output as the oracle. (Until round 4 there
    were two kernel variants and this test forced each onto the other's shapes with AIM_GA_LONG; the banded scan path now serves every window
    and the variable is ignored.)"""
    _hip_vs_oracle(l, err, n, 300 + l, True)
    _hip_vs_oracle(l, err, min(n, 100), 400 + l, False)
    _hip_vs_oracle(l, err, min(n, 64), 500 + l, True)


@pytest.mark.gpu
@pytest.mark.parametrize("poison", [None, "165", "255"])
def test_genasm_banded_sweep_at_the_band_edge(gpu, monkeypatch, poison):
    """Round 4: the fast path keeps 32-bit banded words (diagonals -15 .. +16) instead of 64-bit vectors. Error rates of
    15-25 % put windows at 10-15 edits -- alignments that reach the band's edge -- and just beyond (the 64-level path); indel-only edits drift
    to one side of the band. Also run with LDS poisoned at kernel entry: the sweep hands masks and columns between lanes through LDS, and a
    result that depends on what LDS held before is a missing ordering (found by tools/fuzz_parity.py --focus genasm under
    AIM_DEBUG_POISON_LDS)."""
    from aim_amd import engine
    from oracle import oracle
    if poison: monkeypatch.setenv("AIM_DEBUG_POISON_LDS", poison)
    for l, err, n, seed in ((1000, 0.2, 625, 11), (700, 0.25, 300, 12), (2000, 0.15, 200, 13)):
        rs = ((int(l * (1 + err)) + 8 + 7) // 8) * 8
        req, pat, txt = engine.gen_pairs(seed, 0, n, l, err, rs)
        rng = np.random.RandomState(seed)
        for i in range(0, n, 5):     # deletions only: the text is the pattern with ~err of its bases dropped (the walk drifts to one side)
            pl = int(req["pattern_len"][i])
            keep = rng.rand(pl) >= err
            t = pat[i, :pl][keep]
            txt[i, :] = 0
            txt[i, : len(t)] = t
            req["text_len"][i] = len(t)
        params = engine.make_params("genasm", 0, rs, backtrace=True)
        res, ops = engine.align(params, req, pat, txt)
        ores, oops, worst = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), req["pattern_len"], req["text_len"], pat, txt, nthreads=16)
        assert worst == 0
        for f in ("score", "end_offset", "status"):
            assert np.array_equal(res[f], ores[f]), f
        for i in range(n):
            e = int(res["end_offset"][i])
            assert np.array_equal(ops[i, :e], oops[i, :e]), i
        _check_alignment(req, pat, txt, res, ops)


@pytest.mark.gpu
def test_genasm_compact_cigar_beyond_65535_runs_reports_overflow(gpu, tmp_path):
    """aim_cigar_t.n_runs is 16 bits. A long pair whose alignment alternates M / X has more runs than that: the device-side
    run-length encoder must flag AIM_CIGAR_OVERFLOW (aim_set_wait: AIM_ENOMEM) instead of storing a truncated count -- and the host CLI
    then prints that batch from result_t + ops rows like the reference, i.e. the oracle's text."""
    import subprocess, sys
    from conftest import ROOT
    from aim_amd import capi, engine
    from oracle import oracle
    l = 110000               # (windows of > 15 edits drift into indels: ~0.64 runs per base)
    rs = ((l + 8 + 7) // 8) * 8
    rng = np.random.RandomState(5)
    req = np.zeros(3, dtype=engine.REQUEST_DTYPE)
    pat = np.zeros((3, rs), dtype=np.uint8); txt = np.zeros((3, rs), dtype=np.uint8)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(3):
        p = rng.choice(bases, size=l)
        t = p.copy()
        if i == 1:
            t[1::2] = bases[(np.searchsorted(bases, p[1::2]) + 1) % 4]          # every second base substituted: M X M X ... = ~70 000 runs in all
        else:
            t[100 * (i + 1)] = bases[(np.searchsorted(bases, p[100 * (i + 1)]) + 1) % 4]
        pat[i, :l], txt[i, :l] = p, t
        req[i]["pattern_len"] = req[i]["text_len"] = l
        req[i]["idx"] = i
    params = engine.make_params("genasm", 0, rs, backtrace=True)
    res, ops = engine.align(params, req, pat, txt)
    o = ops[1, : int(res["end_offset"][1])]
    assert 1 + int((o[1:] != o[:-1]).sum()) > 65535                          # the case is what it claims to be
    with engine.DeviceSet(1) as s:
        s.configure_slots(params, 3, slots=1, max_raw=0, max_runs=400000)
        s.submit(0, 0, req, pat=pat, txt=txt, cigar_runs_cap=400000)
        io, keep, out = s._inflight[(0, 0)]
        rc = s.lib.aim_set_wait(s.handle, 0, 0, None)
        s._inflight.pop((0, 0))
        assert rc == capi.AIM_ENOMEM, rc
        cig = out["cig"]
        assert int(cig["status"][1]) & capi.CIGAR_OVERFLOW and int(cig["n_runs"][1]) == 0
        assert int(cig["status"][0]) == 0 and int(cig["n_runs"][0]) == 3 and int(cig["status"][2]) == 0
    inp = tmp_path / "in.seq"
    inp.write_bytes(engine.pairs_to_text(req, pat, txt))
    out_f = tmp_path / "out"
    r = subprocess.run([os.path.join(ROOT, "aim_amd", "host", "host"), str(inp), str(out_f), "3", "--algo", "genasm", "--read-size", str(rs), "--backtrace"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    ores, oops, _ = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), req["pattern_len"], req["text_len"], pat, txt, nthreads=3)
    ores["idx"] = req["idx"]
    assert out_f.read_bytes() == oracle.format_output(ores, oops, True)
