"""Parity tests proper: the HIP path, called through the C-ABI (aim_set_* via aim_amd.engine.DeviceSet),
against the CPU oracle on the same inputs -- bit-exact scores, offsets and CIGAR ops -- and against the
reference digests recorded for Datasets/sample-l100-e1-40K.  Run on the GPU box with `-m gpu`."""
import collections
import os

import numpy as np
import pytest

from conftest import (ROOT, judge_abort_cases, judge_case_input, judge_cases, judge_costs, judge_dataset_cases, md5,
                      reduce_cases)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(built):
    from aim_amd import capi
    import ctypes as C
    lib = capi.load()
    n = C.c_int()
    rc = lib.aim_device_count(C.byref(n))
    assert rc == 0 and n.value >= 1, "no HIP device visible: %s" % lib.aim_last_error()
    return lib


def _oracle_params(oracle, params, algo):
    from aim_amd import capi
    bt = bool(params.flags & capi.FLAG_BACKTRACE)
    red = bool(params.flags & capi.FLAG_REDUCE)
    cellb = 2 if (params.flags & capi.FLAG_SWG_W16) else 0
    return oracle.params(algo, params.max_score, params.read_size, match=params.match, mismatch=params.mismatch,
                         gap_o=params.gap_o, gap_e=params.gap_e, gap_i=params.gap_i, gap_d=params.gap_d, backtrace=bt, reduce=red,
                         swg_cell_bytes=cellb)


def _compare(algo, params, req, pat, txt, threads=8, expect_status=None, env=None):
    """Align on the GPU and with the oracle; require bit-identical results (and ops inside [begin,end))."""
    from aim_amd import capi, engine
    from oracle import oracle
    res, ops = engine.align(params, req, pat, txt, check=False)
    op = _oracle_params(oracle, params, algo)
    ores, oops, worst = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=threads)
    assert np.array_equal(res["idx"], req["idx"])
    for f in ("score", "max_operations", "end_offset", "status"):
        bad = np.nonzero(res[f] != ores[f])[0]
        assert bad.size == 0, "%s differs at pair %d: hip %d oracle %d (plen %d tlen %d)" % (
            f, bad[0], res[f][bad[0]], ores[f][bad[0]], req["pattern_len"][bad[0]], req["text_len"][bad[0]])
    if params.flags & capi.FLAG_BACKTRACE:
        ok = res["status"] == 0
        bad = np.nonzero((res["begin_offset"] != ores["begin_offset"]) & ok)[0]
        assert bad.size == 0, "begin_offset differs at pair %d: hip %d oracle %d" % (
            bad[0], res["begin_offset"][bad[0]], ores["begin_offset"][bad[0]])
        for i in np.nonzero(ok)[0]:
            b, e = int(res["begin_offset"][i]), int(res["end_offset"][i])
            if not np.array_equal(ops[i, b:e], oops[i, b:e]):
                raise AssertionError("ops differ at pair %d: hip %r oracle %r" % (
                    i, ops[i, b:e].tobytes(), oops[i, b:e].tobytes()))
    if expect_status is not None:
        assert collections.Counter(int(s) for s in res["status"]) == expect_status
    return res, ops, ores


# ------------------------------------------------------------------ reference digests on the sample file
SAMPLE_CASES = [
    ("wfa_backtrace", "wfa", 5, dict(backtrace=True)),
    ("wfa_reduce_backtrace", "wfa", 5, dict(backtrace=True, reduce=True)),
    ("wfa_score_only", "wfa", 5, dict(reduce=True)),
    ("nw_backtrace", "nw", 4, dict(backtrace=True)),
    ("swg_w8_backtrace", "swg", 5, dict(backtrace=True)),
    ("swg_w16_backtrace", "swg", 5, dict(backtrace=True, swg_w16=True)),
]


@pytest.mark.parametrize("key,algo,ms,kw", SAMPLE_CASES)
def test_sample_file_matches_reference_digest(gpu, sample_bytes, ref_digests, key, algo, ms, kw):
    from aim_amd import engine
    req, pat, txt = engine.parse_pairs(sample_bytes, 112)
    params = engine.make_params(algo, ms, 112, **kw)
    res, ops = engine.align(params, req, pat, txt)
    out = engine.format_output(res, ops, kw.get("backtrace", False))
    assert md5(out) == ref_digests[key]


@pytest.mark.parametrize("case", judge_cases(), ids=lambda c: c["name"])
def test_judge_r01_reference_digests_through_hip(gpu, case):
    """The wider reference digests of judge_r01 / r02 / r03_cases.json, through aim_set_* on the GPU."""
    from aim_amd import engine
    data = judge_case_input(case)
    req, pat, txt = engine.parse_pairs(data, case["read_size"])
    params = engine.make_params(case["algo"], case["max_score"], case["read_size"], backtrace=case["backtrace"],
                                reduce=case.get("reduce", False), swg_w16=case.get("swg_cell_bytes", 0) == 2, **judge_costs(case))
    res, ops = engine.align(params, req, pat, txt)
    assert md5(engine.format_output(res, ops, case["backtrace"])) == case["output_md5"]


def _host_cli(case, inp, out, cwd, extra=()):
    import subprocess
    c = judge_costs(case)
    cmd = [os.path.join(ROOT, "aim_amd", "host", "host"), str(inp), str(out), str(case.get("n", case.get("gen", {}).get("n"))),
           "--algo", case["algo"], "--max-score", str(case["max_score"]), "--read-size", str(case["read_size"]),
           "--nr-dpus", str(case.get("nr_dpus", 1)), "--mismatch", str(c.get("mismatch", 3)), "--gap-o", str(c.get("gap_o", 4)),
           "--gap-e", str(c.get("gap_e", 1)), "--gap", str(c.get("gap", 4))]
    cmd += sum(([f, str(c[k])] for f, k in (("--gap-i", "gap_i"), ("--gap-d", "gap_d")) if k in c), [])
    cmd += (["--backtrace"] if case["backtrace"] else []) + (["--reduce"] if case.get("reduce") else [])
    cmd += ["--swg-w16"] if case.get("swg_cell_bytes", 0) == 2 else []
    return subprocess.run(cmd + list(extra), capture_output=True, text=True, cwd=str(cwd))


@pytest.mark.parametrize("case", judge_abort_cases(), ids=lambda c: c["name"])
def test_judge_r03_abort_cases_through_hip(gpu, case, tmp_path):
    """judge r03: where the reference stops with `SWG backtrace. No backtrace operation found` + exit(1) (swg.c:99-104) the HIP path
    reports AIM_PAIR_SWG_NO_OP for exactly the pairs the oracle does, and the CLI prints the message and exits 1 -- on every
    output path (compact CIGAR, ops rows, ASCII rows)."""
    from aim_amd import capi, engine
    data = judge_case_input(case)
    req, pat, txt = engine.parse_pairs(data, case["read_size"])
    params = engine.make_params(case["algo"], case["max_score"], case["read_size"], backtrace=True, **judge_costs(case))
    res, _, ores = _compare(case["algo"], params, req, pat, txt)
    assert (res["status"] == capi.PAIR_SWG_NO_OP).any() and set(res["status"].tolist()) <= {0, capi.PAIR_SWG_NO_OP}
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(data)
    for extra in ((), ("--full-ops",), ("--no-pack", "--full-ops")):
        r = _host_cli(case, inp, out, tmp_path, extra)
        assert r.returncode == 1 and case["abort"] in r.stdout, (extra, r.stdout, r.stderr)


@pytest.mark.parametrize("case", judge_dataset_cases(), ids=lambda c: c["name"])
def test_judge_r03_dataset_digests_through_the_host_cli(gpu, case, err_full_bytes, tmp_path):
    """judge r03: whole-file digests of the reference on its own Datasets/ERR240727-l100-e1-30000Pairs (real reads with 'N':
    the raw side list), NR_DPUS 4, through the drop-in CLI -- packed + compact output, and the reference's own wire formats."""
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(err_full_bytes)
    for extra in ((), ("--no-pack", "--full-ops") if case["backtrace"] else ("--no-pack",), ("--batch", "4000", "--threads", "5")):
        r = _host_cli(case, inp, out, tmp_path, extra)
        assert r.returncode == 0, (extra, r.stdout, r.stderr)
        assert md5(out.read_bytes()) == case["output_md5"], extra


@pytest.mark.parametrize("case", [c for c in judge_cases() if c["name"].startswith("tails_") and c["gen"]["l"] == 100],
                         ids=lambda c: c["name"])
def test_judge_r05_tail_heavy_digests_through_the_host_cli(gpu, case, tmp_path):
    """judge r05: the three l = 100 `tails_v1` inputs (a third of the pairs with plen >= tlen + 2, some with plen > 2 tlen: tail
    cells in the register kernels, the to-do passes, the literal cells) through the drop-in CLI on both wire formats."""
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(judge_case_input(case))
    for extra in ((), ("--no-pack", "--full-ops") if case["backtrace"] else ("--no-pack",), ("--batch", "700", "--threads", "3")):
        r = _host_cli(case, inp, out, tmp_path, extra)
        assert r.returncode == 0, (extra, r.stdout, r.stderr)
        assert md5(out.read_bytes()) == case["output_md5"], extra


def test_sample_file_wave_kernel_too(gpu, sample_bytes, ref_digests, monkeypatch):
    """The general one-pair-per-wavefront WFA kernel must give the same file (fast path disabled)."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_FORCE_WAVE", "1")
    req, pat, txt = engine.parse_pairs(sample_bytes, 112)
    for key, kw in (("wfa_reduce_backtrace", dict(backtrace=True, reduce=True)), ("wfa_score_only", dict(reduce=True))):
        params = engine.make_params("wfa", 5, 112, **kw)
        res, ops = engine.align(params, req, pat, txt)
        assert md5(engine.format_output(res, ops, kw.get("backtrace", False))) == ref_digests[key]


def test_real_reads_with_N(gpu, err_bytes):
    from aim_amd import engine
    req, pat, txt = engine.parse_pairs(err_bytes, 112)
    assert len(req) == 2000
    for algo, ms, kw in (("wfa", 5, dict(backtrace=True, reduce=True)), ("wfa", 5, dict()), ("nw", 4, dict(backtrace=True)),
                         ("swg", 5, dict(backtrace=True))):
        _compare(algo, engine.make_params(algo, ms, 112, **kw), req, pat, txt)


# ------------------------------------------------------------------ synthetic strata
@pytest.mark.parametrize("err", [0.01, 0.02, 0.05, 0.10])
@pytest.mark.parametrize("algo,kw", [("wfa", dict()), ("wfa", dict(backtrace=True)), ("wfa", dict(backtrace=True, reduce=True)),
                                     ("nw", dict(backtrace=True)), ("swg", dict(backtrace=True)),
                                     ("swg", dict(backtrace=True, swg_w16=True)), ("swg", dict())])
def test_synthetic_l100(gpu, algo, kw, err):
    from aim_amd import engine
    ms, rs = engine.launcher_sizes(algo, 100, err)
    req, pat, txt = engine.gen_pairs(1234 + int(err * 100), 0, 3000, 100, err, rs)
    strata = collections.Counter(np.sign(req["pattern_len"] - req["text_len"]).tolist())
    assert len(strata) == 3   # plen > tlen (flat-index aliasing N1/S1), plen < tlen, plen == tlen all present
    _compare(algo, engine.make_params(algo, ms, rs, **kw), req, pat, txt)


@pytest.mark.parametrize("force_wave", ["0", "1"])
def test_wfa_custom_penalties_and_overflow(gpu, force_wave, monkeypatch):
    """Non-default costs; MAX_SCORE deliberately too small so that the score cap (W3) triggers."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_FORCE_WAVE", force_wave)
    for (x, o, e), err, ms_override in (((4, 6, 2), 0.05, None), ((2, 3, 1), 0.05, None), ((5, 4, 2), 0.03, None),
                                        ((3, 4, 1), 0.05, 7), ((1, 1, 1), 0.04, None),
                                        # gcd(x, o+e, e) > 1: wfa_group_kernel counts in score units (GroupCfg::unit) -- caps on both sides of a multiple
                                        ((4, 6, 2), 0.05, 9), ((4, 6, 2), 0.05, 10), ((2, 4, 2), 0.05, None), ((6, 3, 3), 0.04, 14), ((6, 3, 3), 0.04, None)):
        ms, rs = engine.launcher_sizes("wfa", 100, err, mismatch=x, gap_o=o, gap_e=e)
        if ms_override is not None:
            ms = ms_override
        req, pat, txt = engine.gen_pairs(99, 0, 1500, 100, err, rs)
        for kw in (dict(), dict(backtrace=True, reduce=True)):
            p = engine.make_params("wfa", ms, rs, mismatch=x, gap_o=o, gap_e=e, **kw)
            res, _, _ = _compare("wfa", p, req, pat, txt)
            if ms_override is not None:
                assert (res["score"] == ms + 1).any()


def test_wfa_adaptive_l1000_e5_backtrace(gpu):
    """BASELINE config 3 shape (MAX_SCORE 250, READ_SIZE 1064): reduce fires (width >= 10)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("wfa", 1000, 0.05)
    assert (ms, rs) == (250, 1064)
    req, pat, txt = engine.gen_pairs(7, 0, 600, 1000, 0.05, rs)
    r1, _, _ = _compare("wfa", engine.make_params("wfa", ms, rs, backtrace=True, reduce=True), req, pat, txt)
    r2, _, _ = _compare("wfa", engine.make_params("wfa", ms, rs, backtrace=True), req, pat, txt)
    _compare("wfa", engine.make_params("wfa", ms, rs, reduce=True), req, pat, txt)
    assert r1["score"].mean() > 100


def test_nw_swg_l250(gpu):
    from aim_amd import engine
    for algo in ("nw", "swg"):
        ms, rs = engine.launcher_sizes(algo, 250, 0.05)
        req, pat, txt = engine.gen_pairs(5, 0, 400, 250, 0.05, rs)
        _compare(algo, engine.make_params(algo, ms, rs, backtrace=True), req, pat, txt)


def test_swg_int8_wrap(gpu):
    """l=150 e=3%: MAX_SCORE 23 < 127 -> int8 cells whose boundary values (4+v) wrap past 127 (S3)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("swg", 150, 0.03)
    assert ms < 127
    req, pat, txt = engine.gen_pairs(11, 0, 500, 150, 0.03, rs)
    _compare("swg", engine.make_params("swg", ms, rs), req, pat, txt)
    _compare("swg", engine.make_params("swg", ms, rs, backtrace=True), req, pat, txt)


# ------------------------------------------------------------------ edge cases
def _mk(pairs, rs):
    from aim_amd import capi
    n = len(pairs)
    req = np.zeros(n, dtype=capi.REQUEST_DTYPE)
    pat = np.zeros((n, rs), dtype=np.uint8)
    txt = np.zeros((n, rs), dtype=np.uint8)
    for i, (p, t) in enumerate(pairs):
        pat[i, : len(p)] = np.frombuffer(p, dtype=np.uint8)
        txt[i, : len(t)] = np.frombuffer(t, dtype=np.uint8)
        req[i] = (len(p), len(t), 0, 1000 + i)
    return req, pat, txt


EDGE_PAIRS = [
    (b"", b""), (b"A", b"A"), (b"A", b"C"), (b"", b"ACGT"), (b"ACGT", b""), (b"ACGTACGT", b"ACGTACGT"),
    (b"ACGTACGTAC", b"ACGTTCGTAC"), (b"AAAAAAAAAA", b"AAAAAAAAAAAA"), (b"AAAAAAAAAAAA", b"AAAAAAAAAA"),
    (b"ACGTNNNNACGT", b"ACGTNNNACGT"), (b"acgtacgt", b"ACGTACGT"), (b"GATTACA" * 16, b"GATTACA" * 16),
    (b"GATTACA" * 15, b"GATTACA" * 7 + b"T" + b"GATTACA" * 8), (b"A" * 112, b"A" * 112), (b"A" * 112, b"C" * 112),
    (b"ACGT" * 28, b"TGCA" * 28), (b"A" * 100, b"A" * 50),
]


@pytest.mark.parametrize("algo,ms,kw", [("wfa", 5, dict()), ("wfa", 5, dict(backtrace=True, reduce=True)),
                                        ("wfa", 40, dict(backtrace=True)), ("nw", 4, dict(backtrace=True)),
                                        ("swg", 5, dict(backtrace=True)), ("swg", 200, dict(backtrace=True)),
                                        ("nw", 4, dict())])
def test_edge_cases(gpu, algo, ms, kw):
    from aim_amd import engine
    req, pat, txt = _mk(EDGE_PAIRS, 112)
    _compare(algo, engine.make_params(algo, ms, 112, **kw), req, pat, txt, threads=1)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_ragged_batch_sizes(gpu, n):
    from aim_amd import engine
    req, pat, txt = engine.gen_pairs(3, 17, n, 100, 0.02, 112)
    for algo, ms, kw in (("wfa", 10, dict()), ("wfa", 10, dict(backtrace=True)), ("nw", 8, dict(backtrace=True))):
        _compare(algo, engine.make_params(algo, ms, 112, **kw), req, pat, txt, threads=1)


def test_empty_batch_and_errors(gpu):
    from aim_amd import capi, engine
    req, pat, txt = engine.gen_pairs(3, 0, 0, 100, 0.02, 112)
    with engine.DeviceSet(1) as s:
        s.configure(engine.make_params("wfa", 5, 112), 16)
        s.push(0, req, pat, txt)
        s.launch()
        res, ops = s.pull(0)
        assert len(res) == 0
        with pytest.raises(capi.AimError):
            s.configure(engine.make_params("wfa", 5, 110), 16)     # read_size not a multiple of 8
        with pytest.raises(capi.AimError):
            s.configure(engine.make_params("wfa", 5, 112, mismatch=0), 16)
        r2, p2, t2 = engine.gen_pairs(3, 0, 8, 100, 0.02, 112)
        r2["pattern_len"][3] = 113                                  # longer than READ_SIZE (host.c:119-123)
        s.configure(engine.make_params("wfa", 5, 112), 16)
        with pytest.raises(capi.AimError):
            s.push(0, r2, p2, t2)


# ------------------------------------------------------------------ full BASELINE size (cfg2) + properties
def test_cfg2_full_size_4M_pairs(gpu):
    """WFA score-only l=100 e=1% on 4M synthetic pairs: bit-exact against the oracle over the whole batch,
    plus the size-independent properties (score alphabet, identical pairs score 0, order/idx preserved)."""
    from aim_amd import engine
    from oracle import oracle
    n = 1 << 22
    ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
    req, pat, txt = engine.gen_pairs(42, 0, n, 100, 0.01, rs)
    params = engine.make_params("wfa", ms, rs, reduce=True)
    res, _ = engine.align(params, req, pat, txt)
    assert np.array_equal(res["idx"], np.arange(n, dtype=np.uint32))
    assert set(np.unique(res["score"]).tolist()) <= {0, 3, 5, 6}
    same = (req["pattern_len"] == req["text_len"]) & (pat == txt).all(axis=1)
    assert (res["score"][same] == 0).all() and same.sum() > 0
    op = oracle.params("wfa", ms, rs, reduce=True)
    ores, _, worst = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=os.cpu_count() or 8)
    assert worst == 0
    assert np.array_equal(res["score"], ores["score"])
    # checksum of checksums: order-sensitive digest of (idx, score)
    assert md5(res["score"].tobytes()) == md5(ores["score"].tobytes())
    # the same 4M pairs through the opt-in compact layouts (8-B WFA request_t in, {idx, score} out): what bench.py times
    cres, _ = engine.align(engine.make_params("wfa", ms, rs, reduce=True, req8=True, res8=True), req, pat, txt)
    assert np.array_equal(cres["idx"], res["idx"]) and np.array_equal(cres["score"], ores["score"])


def test_multi_device_split_matches_single(gpu):
    """Static contiguous split over the devices of a set (host.c:191-209) returns the same results in input order.
    With one physical GPU the set is built from the same device twice."""
    from aim_amd import engine
    req, pat, txt = engine.gen_pairs(8, 0, 5001, 100, 0.02, 112)
    params = engine.make_params("wfa", 10, 112, backtrace=True)
    a_res, a_ops = engine.align(params, req, pat, txt)
    with engine.DeviceSet(device_ids=[0, 0, 0]) as s:
        b_res, b_ops = s.align(params, req, pat, txt)
    assert np.array_equal(a_res, b_res)
    for i in range(len(req)):
        b, e = int(a_res["begin_offset"][i]), int(a_res["end_offset"][i])
        assert np.array_equal(a_ops[i, b:e], b_ops[i, b:e])


def test_host_cli_end_to_end(gpu, sample_bytes, ref_digests, tmp_path):
    """The C host program keeps the reference CLI/output: whole-file digests through `python -m aim_amd.launch`."""
    import subprocess
    import sys
    inp = tmp_path / "sample"
    inp.write_bytes(sample_bytes)
    for algo, flags, key in (("wfa", ["-b", "-r"], "wfa_reduce_backtrace"), ("wfa", ["-r"], "wfa_score_only"),
                             ("nw", ["-b"], "nw_backtrace"), ("swg", ["-b"], "swg_w8_backtrace")):
        out = tmp_path / ("out_" + key)
        r = subprocess.run([sys.executable, "-m", "aim_amd.launch", algo, "-i", str(inp), "-o", str(out), "-l", "100",
                            "-e", "0.01", "-n", "20000", "-d", "4"] + flags, capture_output=True, text=True,
                           cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(__file__))))
        assert r.returncode == 0, r.stdout + r.stderr
        assert md5(out.read_bytes()) == ref_digests[key]
        for line in ("Allocated 4 DPU(s)", "NumReads per dpu = 5000", "DPU Kernel:", "CPU-DPU:", "DPU-CPU:"):
            assert line in r.stdout


def test_fast_path_handles_non_acgt_bytes_itself(gpu, sample_bytes, err_bytes):
    """cfg2 runs on the one-pair-per-lane kernel. Pairs with bytes outside ACGT (the reference compares raw bytes, 'N'
    matches 'N') are aligned by the SAME kernel from the raw rows (round 2: no to-do list, no second launch, no scratch);
    wfa_group_kernel still hands such pairs to the general kernel and reports how many."""
    from aim_amd import capi, engine
    import ctypes as C
    lib = capi.load()
    params = engine.make_params("wfa", 5, 112, reduce=True)
    assert lib.aim_kernel_name(C.byref(params)) == b"wfa_lane_kernel"
    assert lib.aim_scratch_bytes(C.byref(params), 1 << 22) == 256     # a token: the kernel uses no scratch
    gparams = engine.make_params("wfa", 12, 112, reduce=True)      # MAX_SCORE 12 -> wfa_group_kernel
    assert lib.aim_kernel_name(C.byref(gparams)) == b"wfa_group_kernel"
    for data in (sample_bytes, err_bytes):
        req, pat, txt = engine.parse_pairs(data, 112)
        expect = 0
        for i in range(len(req)):
            s = pat[i, : req["pattern_len"][i]].tobytes() + txt[i, : req["text_len"][i]].tobytes()
            expect += 1 if (set(s) - set(b"ACGT")) else 0
        with engine.DeviceSet(1) as ds:
            ds.align(params, req, pat, txt)
            assert ds.fallback_pairs(0) == 0
            assert "wfa_lane_kernel" in ds.plan_describe(0)
        with engine.DeviceSet(1) as ds:
            ds.align(gparams, req, pat, txt)
            assert ds.fallback_pairs(0) == expect
        if data is err_bytes:
            assert expect > 0
        for kw in (dict(reduce=True), dict(reduce=True, backtrace=True)):
            _compare("wfa", engine.make_params("wfa", 5, 112, **kw), req, pat, txt)
    # lower-case / N / arbitrary bytes (incl. >= 0x80 and NUL inside a read), mixed into an otherwise clean batch;
    # equal non-ACGT bytes on both sides must MATCH (raw byte compare), different ones must not
    req, pat, txt = engine.gen_pairs(77, 0, 4000, 100, 0.01, 112)
    for i in range(0, 4000, 7):
        pat[i, i % 100] = ord("N")
    for i in range(3, 4000, 11):
        txt[i, (3 * i) % 99] = ord("a")
    for i in range(5, 4000, 13):
        j = (5 * i) % 95
        pat[i, j] = 0xC1 if i % 2 else 0x00
        txt[i, j] = 0xC1 if i % 2 else 0x00
    for i in range(1, 4000, 17):
        pat[i, 7] = ord("n"); txt[i, 7] = ord("N")
    for rs_kw in (dict(reduce=True), dict(reduce=True, backtrace=True), dict()):
        _compare("wfa", engine.make_params("wfa", 5, 112, **rs_kw), req, pat, txt)
    # same at READ_SIZE 80 (the other lane shape)
    req, pat, txt = engine.gen_pairs(78, 0, 2000, 64, 0.02, 80)
    for i in range(0, 2000, 5):
        pat[i, i % 60] = ord("N"); txt[i, (i + 1) % 60] = ord("N")
    for kw in (dict(reduce=True), dict(backtrace=True)):
        _compare("wfa", engine.make_params("wfa", 5, 80, **kw), req, pat, txt)


# ------------------------------------------------------------------ long-read NW/SWG kernels (dp_strip: column-strip pipeline; dp_wave: row scan)
DP_KERNELS = [dict(), dict(AIM_DPW_LEGACY="1"), dict(AIM_STRIP_K="32"), dict(AIM_STRIP_K="20")]


def _dp_kernel_name(env, params):
    from aim_amd import capi
    int8 = params.algo == capi.ALGO_SWG and params.max_score < 127 and not (params.flags & capi.FLAG_SWG_W16)
    bt = bool(params.flags & capi.FLAG_BACKTRACE)
    nw = params.algo == capi.ALGO_NW
    rs_ok = params.read_size >= 177 and (params.read_size <= 1024 or ((nw and params.read_size <= 1280) or 1440 <= params.read_size <= (2560 if nw else 2048) if bt else params.read_size <= (1792 if nw else 1536)))
    if not env and not int8 and rs_ok:
        return b"dp_group_kernel"            # round 5: medium reads (the long-read kernels' knobs keep them on dp_strip / dp_wave); round 6: dp_group_rs_ok's ranges
    if int8 and not env and params.read_size <= 1199 and not os.environ.get("AIM_FORCE_DPWAVE"):
        return b"swg_lane_kernel"            # round 6: int8 cells on one pair per lane while 64 lanes' rows fit LDS
    return b"dp_wave_kernel" if (env.get("AIM_DPW_LEGACY") or int8) else b"dp_strip_kernel"


@pytest.mark.parametrize("env", DP_KERNELS)
@pytest.mark.parametrize("key,algo,ms,kw", [("nw_backtrace", "nw", 4, dict(backtrace=True)),
                                            ("swg_w16_backtrace", "swg", 5, dict(backtrace=True, swg_w16=True)),
                                            ("swg_w8_backtrace", "swg", 5, dict(backtrace=True))])
def test_dp_wave_on_sample_file_digest(gpu, sample_bytes, ref_digests, monkeypatch, key, algo, ms, kw, env):
    """The long-read kernels (strip pipeline, row scan; for int8 cells the literal path) forced onto the reference's sample file."""
    from aim_amd import capi, engine
    import ctypes as C
    monkeypatch.setenv("AIM_FORCE_DPWAVE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    req, pat, txt = engine.parse_pairs(sample_bytes, 112)
    n = 20000 if "w8" not in key else 4000
    params = engine.make_params(algo, ms, 112, **kw)
    assert capi.load().aim_kernel_name(C.byref(params)) == _dp_kernel_name(env, params)
    if n == 20000:
        res, ops = engine.align(params, req, pat, txt)
        assert md5(engine.format_output(res, ops, True)) == ref_digests[key]
    else:
        _compare(algo, params, req[:n], pat[:n], txt[:n])


@pytest.mark.parametrize("env", DP_KERNELS)
@pytest.mark.parametrize("algo,kw", [("nw", dict(backtrace=True)), ("swg", dict(backtrace=True, swg_w16=True)), ("swg", dict(swg_w16=True))])
@pytest.mark.parametrize("err", [0.02, 0.10])
def test_dp_wave_strata_l100(gpu, monkeypatch, algo, kw, err, env):
    """plen > tlen (tail cells / boundary aliasing), plen < tlen and plen == tlen through the long-read kernels."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_FORCE_DPWAVE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ms, rs = engine.launcher_sizes(algo, 100, err)
    req, pat, txt = engine.gen_pairs(4321, 0, 2000, 100, err, rs)
    assert (req["pattern_len"] > req["text_len"] + (2 if err > 0.05 else 0)).any()
    _compare(algo, engine.make_params(algo, ms, rs, **kw), req, pat, txt)


@pytest.mark.parametrize("env", DP_KERNELS)
@pytest.mark.parametrize("algo", ["nw", "swg"])
@pytest.mark.parametrize("l,err,n", [(600, 0.05, 300), (1000, 0.05, 200), (3000, 0.02, 24), (1500, 0.03, 700)])
def test_dp_wave_long_reads(gpu, monkeypatch, algo, l, err, n, env):
    from aim_amd import capi, engine
    import ctypes as C
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ms, rs = engine.launcher_sizes(algo, l, err)
    req, pat, txt = engine.gen_pairs(l, 0, n, l, err, rs)
    params = engine.make_params(algo, ms, rs, backtrace=True)
    assert capi.load().aim_kernel_name(C.byref(params)) == _dp_kernel_name(env, params)
    _compare(algo, params, req, pat, txt)
    _compare(algo, engine.make_params(algo, ms, rs), req, pat, txt)


def test_cfg4_swg_l10000_e1(gpu):
    """BASELINE config 4 shape: SWG, l = 10 000, e = 1 % (MAX_SCORE 500 -> int16 cells, READ_SIZE 10112)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("swg", 10000, 0.01)
    assert (ms, rs) == (500, 10112)
    req, pat, txt = engine.gen_pairs(10, 0, 8, 10000, 0.01, rs)
    assert (req["pattern_len"] > req["text_len"]).any() and (req["pattern_len"] < req["text_len"]).any()
    res, _, _ = _compare("swg", engine.make_params("swg", ms, rs, backtrace=True), req, pat, txt, threads=4)
    assert (res["score"] > 100).all()


def test_dp_wave_exotic_and_literal_paths(gpu, monkeypatch):
    """plen > 2*tlen (tail would spill past the next row) and configurations where an int16 store could wrap
    take the literal single-lane path; both must still equal the oracle."""
    from aim_amd import engine
    rng = np.random.default_rng(5)
    pairs = []
    for _ in range(40):
        pl, tl = int(rng.integers(1, 330)), int(rng.integers(1, 120))
        p = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=pl).tolist())
        t = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=tl).tolist())
        pairs.append((p, t))
    pairs += [(b"", b""), (b"ACGT" * 80, b"A"), (b"A", b"ACGT" * 80), (b"ACGT" * 60, b"ACGT" * 20), (b"", b"ACGT"), (b"ACGT", b"")]
    req, pat, txt = _mk(pairs, 336)
    assert (req["pattern_len"] > 2 * req["text_len"]).any()
    for algo, ms, kw in (("nw", 40, dict(backtrace=True)), ("swg", 200, dict(backtrace=True)), ("swg", 40, dict(backtrace=True)),
                         ("nw", 40, dict(backtrace=True, gap=50))):
        _compare(algo, engine.make_params(algo, ms, 336, **kw), req, pat, txt, threads=2)


@pytest.mark.parametrize("env", [dict(AIM_WFA_NO_RING="1"), dict(AIM_WFA_SLOTW="16"), dict(AIM_WFA_SLOTW="64")])
def test_wfa_wave_storage_modes(gpu, monkeypatch, env):
    """The general WFA kernel keeps the live wavefront window in an LDS ring; wavefronts wider than a slot (forced
    here with tiny slots) and the ring-less mode live in the HBM pool.  All modes must give the same results."""
    from aim_amd import engine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("AIM_FORCE_WAVE", "1")
    ms, rs = engine.launcher_sizes("wfa", 1000, 0.05)
    req, pat, txt = engine.gen_pairs(21, 0, 200, 1000, 0.05, rs)
    for kw in (dict(backtrace=True, reduce=True), dict(backtrace=True), dict(reduce=True), dict()):
        _compare("wfa", engine.make_params("wfa", ms, rs, **kw), req, pat, txt)
    ms, rs = engine.launcher_sizes("wfa", 100, 0.10)
    req, pat, txt = engine.gen_pairs(22, 0, 1500, 100, 0.10, rs)
    for kw in (dict(backtrace=True, reduce=True), dict()):
        _compare("wfa", engine.make_params("wfa", ms, rs, **kw), req, pat, txt)


# ------------------------------------------------------------------ BASELINE full sizes: properties + bit-exactness
def _cigar_properties(req, pat, txt, res, ops, mismatch, gap_o, gap_e, check_score):
    """Size-independent properties of a (score, CIGAR) answer: the ops consume exactly the pattern (M/X/D) and the text
    (M/X/I), 'M' only over equal bases and 'X' only over different ones, and the affine cost of the CIGAR equals the score."""
    n = len(req)
    for i in range(n):
        b, e = int(res["begin_offset"][i]), int(res["end_offset"][i])
        o = ops[i, b:e]
        pl, tl = int(req["pattern_len"][i]), int(req["text_len"][i])
        is_m, is_x, is_i, is_d = (o == ord("M")), (o == ord("X")), (o == ord("I")), (o == ord("D"))
        assert (is_m | is_x | is_i | is_d).all()
        adv_p = (is_m | is_x | is_d).astype(np.int64)
        adv_t = (is_m | is_x | is_i).astype(np.int64)
        assert adv_p.sum() == pl and adv_t.sum() == tl, i
        pi = np.cumsum(adv_p) - adv_p
        ti = np.cumsum(adv_t) - adv_t
        diag = is_m | is_x
        eq = pat[i, pi[diag]] == txt[i, ti[diag]]
        assert (eq == is_m[diag]).all(), i
        if check_score:
            gap = is_i | is_d
            kind = np.where(is_i, 1, np.where(is_d, 2, 0))
            opens = gap & np.concatenate(([True], kind[1:] != kind[:-1]))
            cost = mismatch * int(is_x.sum()) + gap_e * int(gap.sum()) + gap_o * int(opens.sum())
            assert cost == int(res["score"][i]), (i, cost, int(res["score"][i]))


def test_cfg3_full_size_262144_pairs(gpu):
    """BASELINE config 3 at full size: WFA-adaptive with CIGAR, l=1000, e=5%, 262 144 pairs -- bit-exact against the
    oracle over the whole batch (scores, offsets) plus CIGAR properties on a sample."""
    from aim_amd import engine
    from oracle import oracle
    n = 1 << 18
    ms, rs = engine.launcher_sizes("wfa", 1000, 0.05)
    req, pat, txt = engine.gen_pairs(3, 0, n, 1000, 0.05, rs)
    params = engine.make_params("wfa", ms, rs, backtrace=True, reduce=True)
    res, ops = engine.align(params, req, pat, txt)
    op = oracle.params("wfa", ms, rs, backtrace=True, reduce=True)
    ores, oops, worst = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=min(64, os.cpu_count() or 8))
    assert worst == 0
    for f in ("score", "begin_offset", "end_offset", "max_operations"):
        assert np.array_equal(res[f], ores[f]), f
    # every CIGAR byte inside [begin, end): rows are compared with a mask
    col = np.arange(ops.shape[1])[None, :]
    inside = (col >= res["begin_offset"][:, None]) & (col < res["end_offset"][:, None])
    assert np.array_equal(np.where(inside, ops, 0), np.where(inside, oops, 0))
    _cigar_properties(req[:512], pat[:512], txt[:512], res[:512], ops[:512], 3, 4, 1, check_score=True)


def test_cfg4_full_size_1024_pairs_properties(gpu, monkeypatch):
    """BASELINE config 4 at full size: SWG with CIGAR, l=10 000, e=1%, 1024 pairs.  The CPU oracle needs 0.6 GB and
    ~0.5 s per pair, so the whole batch is checked through properties and a 48-pair subset bit-exactly."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_SCRATCH_GB", "64")
    n = 1024
    ms, rs = engine.launcher_sizes("swg", 10000, 0.01)
    req, pat, txt = engine.gen_pairs(4, 0, n, 10000, 0.01, rs)
    params = engine.make_params("swg", ms, rs, backtrace=True)
    res, ops = engine.align(params, req, pat, txt)
    assert np.array_equal(res["idx"], np.arange(n, dtype=np.uint32)) and (res["status"] == 0).all()
    # the alignment properties only hold where the reference's flat-index aliasing (S1) is not in play: with
    # plen > tlen the reference itself emits CIGARs that put 'M' over different bases (checked against the oracle)
    ok = req["pattern_len"] <= req["text_len"]
    assert 200 < ok.sum() < 900
    _cigar_properties(req[ok], pat[ok], txt[ok], res[ok], ops[ok], 3, 4, 1, check_score=True)
    for i in np.nonzero(~ok)[0]:
        o = ops[i, res["begin_offset"][i]:res["end_offset"][i]]
        assert np.isin(o, np.frombuffer(b"MXID", dtype=np.uint8)).all()
    sub = slice(0, 48)
    _compare("swg", params, req[sub], pat[sub], txt[sub], threads=8)


def test_wfa_group_kernel_coverage(gpu, monkeypatch):
    """The G-lanes-per-pair kernel: every group width (G = 1..16), WFA-adaptive on/off with wavefronts wide enough
    for the reduction to fire, custom penalties, non-ACGT fallback, ragged tails."""
    from aim_amd import capi, engine
    import ctypes as C
    lib = capi.load()
    monkeypatch.setenv("AIM_NO_LANE_EXT", "1")   # (l = 100, e = 2 %, score-only) otherwise runs on wfa_lane_kernel's dynamic-bounds shape
    cases = [(100, 0.02, {}), (100, 0.05, {}), (100, 0.10, {}), (150, 0.05, {}), (250, 0.05, {}), (250, 0.08, {}),
             (100, 0.05, dict(mismatch=2, gap_o=3, gap_e=1)), (100, 0.04, dict(mismatch=5, gap_o=4, gap_e=2)),
             (200, 0.03, dict(mismatch=1, gap_o=1, gap_e=1))]
    seen_g = set()
    for l, e, cost in cases:
        ms, rs = engine.launcher_sizes("wfa", l, e, **cost)
        req, pat, txt = engine.gen_pairs(31 + l, 0, 1203, l, e, rs)
        pat[5, 3] = ord("N")
        txt[77, 10] = ord("n")
        for red, bt in ((True, False), (False, False), (True, True), (False, True)):
            params = engine.make_params("wfa", ms, rs, reduce=red, backtrace=bt, **cost)
            assert lib.aim_kernel_name(C.byref(params)) == b"wfa_group_kernel", (l, e, ms, rs)
            with engine.DeviceSet(1) as ds:
                res, _ = ds.align(params, req, pat, txt)
                assert ds.fallback_pairs(0) == 2
            _compare("wfa", params, req, pat, txt)
        seen_g.add((ms, rs))
    assert len(seen_g) >= 6


def test_dp_lane_all_length_relations(gpu):
    """Short-read NW/SWG over every plen/tlen relation, including plen >= 2*(tlen+1) where a tail cell lands two
    rows further down the flat table: the in-place LDS rows and the last-writer mapping must still equal the oracle."""
    from aim_amd import capi, engine
    import ctypes as C
    rng = np.random.default_rng(9)
    pairs = []
    for _ in range(300):
        pl, tl = int(rng.integers(0, 112)), int(rng.integers(0, 112))
        if rng.random() < 0.3:
            tl = int(rng.integers(0, 12))
        p = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=pl).tolist())
        t = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=tl).tolist())
        if rng.random() < 0.5 and pl > 10 and tl > 10:     # related sequences, so real alignments appear too
            t = (p[: tl // 2] + t)[:tl]
        pairs.append((p, t))
    req, pat, txt = _mk(pairs, 112)
    assert (req["pattern_len"] > 2 * req["text_len"] + 2).sum() > 20
    for algo, ms, kw in (("nw", 40, dict(backtrace=True)), ("nw", 40, dict()), ("swg", 40, dict(backtrace=True)),
                         ("swg", 200, dict(backtrace=True)), ("swg", 40, dict())):
        params = engine.make_params(algo, ms, 112, **kw)
        assert capi.load().aim_kernel_name(C.byref(params)) in (b"nw_reg_kernel", b"nw_lane_kernel", b"swg_reg_kernel", b"swg_lane_kernel")   # (nw_reg: nearly all of these pairs take its to-do list to nw_lane_kernel)
        _compare(algo, params, req, pat, txt, threads=4)


def test_host_cli_parser_quirks_match_oracle_cli(gpu, sample_bytes, tmp_path):
    """The parallel mmap parser keeps get_reads' behaviour (host.c:91-134, 191): a last line without newline loses
    its final base, a trailing unpaired line is dropped, n is not a cap (ROUND_UP_8(n/d)*d pairs are consumed) --
    whole-file comparison with the oracle's restatement of the reference host."""
    import subprocess
    from conftest import ROOT
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    cli = os.path.join(ROOT, "oracle", "oracle_cli")
    lines = sample_bytes.split(b"\n")[:2001]                  # 1000 pairs + one unpaired line
    variants = {
        "plain": b"\n".join(lines[:2000]) + b"\n",
        "no_final_newline": b"\n".join(lines[:2000]),         # last text line loses its last base
        "unpaired_tail": b"\n".join(lines[:2001]) + b"\n",
    }
    for name, data in variants.items():
        inp = tmp_path / (name + ".seq")
        inp.write_bytes(data)
        for n, d, flags_h, flags_o in ((1000, 1, ["--backtrace", "--reduce"], ["-b", "-r"]), (100, 4, [], []),
                                       (5000, 3, ["--backtrace"], ["-b"])):
            oh, oo = tmp_path / "h.out", tmp_path / "o.out"
            rh = subprocess.run([host, str(inp), str(oh), str(n), "--algo", "wfa", "--max-score", "5", "--read-size", "112",
                                 "--nr-dpus", str(d), "--threads", "7"] + flags_h, capture_output=True, text=True, cwd=str(tmp_path))
            ro = subprocess.run([cli, "wfa", "-i", str(inp), "-o", str(oo), "-n", str(n), "-l", "100", "-e", "0.01", "-d", str(d)] + flags_o,
                                capture_output=True, text=True)
            assert rh.returncode == 0 and ro.returncode == 0, (name, n, d, rh.stdout, rh.stderr, ro.stderr)
            assert oh.read_bytes() == oo.read_bytes(), (name, n, d)
            assert oh.read_bytes().count(b", \n") == min(len(data.split(b"\n")) // 2 if not data.endswith(b"\n") else data.count(b"\n") // 2,
                                                          ((n // d + 7) // 8) * 8 * d)


@pytest.mark.gpu
def test_wfa_group_lanes_per_pair_plans_agree_with_oracle(gpu, monkeypatch):
    """wfa_group_plan's measured rule: the default plan and the forced AIM_GROUP_G = 16 / 64 plans of the
    mid-range configurations the rule moved (l=400 e=10 % -> G=64) or kept (l=250 e=10 % -> G=16), and a short-window
    configuration forced onto a whole wavefront per pair, all bit-identical to the oracle; fused compute+extend and the
    one-pass reduction are exercised with and without WFA-adaptive and CIGAR; one non-ACGT pair goes to the to-do list."""
    from aim_amd import capi, engine
    import ctypes as C
    lib = capi.load()
    for l, e, n in ((400, 0.10, 300), (250, 0.10, 400), (100, 0.10, 600), (1000, 0.02, 200)):
        ms, rs = engine.launcher_sizes("wfa", l, e)
        req, pat, txt = engine.gen_pairs(1234 + l, 0, n, l, e, rs)
        pat[5, 2] = ord("N")
        for force in (None, "16", "64"):
            if force is None:
                monkeypatch.delenv("AIM_GROUP_G", raising=False)
            else:
                monkeypatch.setenv("AIM_GROUP_G", force)
            for red, bt in ((True, True), (True, False), (False, True)):
                params = engine.make_params("wfa", ms, rs, reduce=red, backtrace=bt)
                assert lib.aim_kernel_name(C.byref(params)) == b"wfa_group_kernel", (l, e, force)
                with engine.DeviceSet(1) as ds:
                    ds.align(params, req, pat, txt)
                    assert ds.fallback_pairs(0) == 1
                _compare("wfa", params, req, pat, txt)
    monkeypatch.delenv("AIM_GROUP_G", raising=False)


@pytest.mark.gpu
def test_dp_wave_every_wavefront_count_agrees_with_oracle(gpu, monkeypatch):
    """dp_wave_nw picks 1 / 2 / 4 wavefronts per pair from the row length and the number of pairs (8 / 10 / 12 for very
    long rows): every count is forced here (AIM_DPW_NW) on long-read NW and SWG with CIGAR, pairs with plen > tlen, = and <
    (tail cell, boundary publishing, tiled traceback), and the default choice for few and for many pairs."""
    from aim_amd import capi, engine
    import ctypes as C
    lib = capi.load()
    for algo, l, e in (("nw", 1000, 0.05), ("swg", 1000, 0.05), ("nw", 2500, 0.02)):
        ms, rs = engine.launcher_sizes(algo, l, e)
        params = engine.make_params(algo, ms, rs, backtrace=True)
        assert lib.aim_kernel_name(C.byref(params)) == (b"dp_group_kernel" if algo == "nw" else b"dp_strip_kernel")   # (round 6: NW with CIGAR at READ_SIZE <= 1280 and 1440 .. 2560 is dp_group_kernel's; AIM_DPW_NW below asks for the strips)
        req, pat, txt = engine.gen_pairs(4321 + l, 0, 96, l, e, rs)
        d = req["pattern_len"].astype(int) - req["text_len"].astype(int)
        assert (d > 0).any() and (d < 0).any()
        for nw in ("1", "2", "4"):
            monkeypatch.setenv("AIM_DPW_NW", nw)
            _compare(algo, params, req, pat, txt)
        monkeypatch.delenv("AIM_DPW_NW", raising=False)
        _compare(algo, params, req, pat, txt)                      # few pairs: the default takes several wavefronts per pair
    ms, rs = engine.launcher_sizes("nw", 1000, 0.05)
    params = engine.make_params("nw", ms, rs, backtrace=True)
    req, pat, txt = engine.gen_pairs(99, 0, 4200, 1000, 0.05, rs)   # > 4096 pairs: the default is one wavefront per pair
    _compare("nw", params, req, pat, txt)


@pytest.mark.gpu
def test_wfa_lane_runtime_max_score_below_shape_with_cigar(gpu):
    """wfa_lane_kernel is compiled for MAX_SCORE <= 5; a smaller run-time MAX_SCORE whose own score has no wavefront (2 and
    4 with penalties 3/4/1) used to let a pair scoring exactly MAX_SCORE+1 be aligned and backtraced instead of reported
    as exceeded (wfa.c:368-376): same score, different begin_offset/CIGAR. Found by tools/fuzz_parity.py."""
    from aim_amd import capi, engine
    import ctypes as C
    lib = capi.load()
    req, pat, txt = engine.gen_pairs(12345, 0, 512, 100, 0.01, 112)
    for ms in (1, 2, 3, 4, 5):
        for bt in (True, False):
            params = engine.make_params("wfa", ms, 112, backtrace=bt, reduce=True)
            assert lib.aim_kernel_name(C.byref(params)) == b"wfa_lane_kernel"
            res, _, ores = _compare("wfa", params, req, pat, txt)
            if ms < 5:
                assert (res["score"] == ms + 1).any()   # the cap is exercised (at 5 nothing in this set exceeds it)


# ------------------------------------------------------------------ round 2: soak, LDS poison, scratch-bound plans
@pytest.mark.parametrize("poison", [None, 255, 0])
def test_dp_wave_soak_launch_to_launch(gpu, poison):
    """Long-read NW/SWG with CIGAR, 9-100 pairs, forced 1/2/4 wavefronts per pair: ~20 s of concurrent repeated launches,
    each compared ON THE DEVICE with the slot's first (oracle-checked) result (tools/soak_dp_wave.py) -- also with every
    workgroup's LDS pre-filled with 0xff / 0x00 (AIM_DEBUG_POISON_LDS): results must not depend on what LDS held."""
    import subprocess, sys, json
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "tools", "soak_dp_wave.py"), "--seconds", "20", "--slots", "6"]
    if poison is not None:
        cmd += ["--poison-lds", str(poison)]
    env = {k: v for k, v in os.environ.items() if not k.startswith("AIM_")}
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["failed_slot"] is None and rep["launches"] > 1000


@pytest.mark.parametrize("poison", ["255", "0"])
def test_every_kernel_is_independent_of_initial_lds(gpu, monkeypatch, poison):
    """AIM_DEBUG_POISON_LDS fills each workgroup's dynamic LDS at kernel entry; all six kernels stay bit-exact."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_DEBUG_POISON_LDS", poison)
    for algo, l, err, n, kw in (("wfa", 100, 0.01, 3000, dict(reduce=True)), ("wfa", 100, 0.01, 3000, dict(reduce=True, backtrace=True)),
                                ("wfa", 100, 0.05, 2000, dict(reduce=True, backtrace=True)), ("wfa", 1000, 0.05, 128, dict(reduce=True, backtrace=True)),
                                ("nw", 100, 0.05, 2000, dict(backtrace=True)), ("swg", 100, 0.05, 2000, dict(backtrace=True)),
                                ("nw", 700, 0.10, 40, dict(backtrace=True)), ("swg", 700, 0.10, 40, dict(backtrace=True))):
        ms, rs = engine.launcher_sizes(algo, l, err)
        req, pat, txt = engine.gen_pairs(77, 0, n, l, err, rs)
        _compare(algo, engine.make_params(algo, ms, rs, **kw), req, pat, txt)
    monkeypatch.setenv("AIM_FORCE_WAVE", "1")
    ms, rs = engine.launcher_sizes("wfa", 100, 0.05)
    req, pat, txt = engine.gen_pairs(78, 0, 1000, 100, 0.05, rs)
    _compare("wfa", engine.make_params("wfa", ms, rs, reduce=True, backtrace=True), req, pat, txt)


def test_wfa_wave_score_only_ring_is_never_shrunk(gpu, monkeypatch):
    """under a small scratch bound the score-only pool (a ring over the live window) used to be shrunk below
    (R+2)*3*(2*MAX_SCORE+3) entries, silently overwriting wavefronts still in use. The plan now drops workgroups instead
    (or fails with AIM_ENOMEM); results stay bit-exact. BACKTRACE keeps the documented AIM_PAIR_NOMEM behaviour."""
    from aim_amd import capi, engine
    monkeypatch.setenv("AIM_SCRATCH_GB", "0.25")
    monkeypatch.setenv("AIM_FORCE_WAVE", "1")
    ms, rs = 5000, 1064
    req, pat, txt = engine.gen_pairs(31, 0, 600, 1000, 0.05, rs)
    params = engine.make_params("wfa", ms, rs, mismatch=9, gap_o=9, gap_e=4, reduce=True)
    res, _, ores = _compare("wfa", params, req, pat, txt)
    assert (res["status"] == 0).all()
    with engine.DeviceSet(1) as s:
        s.configure(params, 600)
        line = s.plan_describe(0)
        cap = int(line.split("pool_cap=")[1].split()[0])
        assert cap >= (max(9, 13) + 2) * 3 * (2 * ms + 3), line


@pytest.mark.parametrize("algo,l,err,kw", [("wfa", 100, 0.01, dict(reduce=True)), ("wfa", 100, 0.05, dict(reduce=True)),
                                           ("wfa", 1000, 0.05, dict(reduce=True)), ("nw", 100, 0.02, dict()),
                                           ("swg", 100, 0.02, dict()), ("nw", 700, 0.05, dict()), ("wfa", 3000, 0.10, dict())])
@pytest.mark.parametrize("req8,res8", [(True, True), (True, False), (False, True)])
def test_compact_io_layouts_match_default(gpu, algo, l, err, kw, req8, res8):
    """AIM_FLAG_REQ8 (the reference's 8-B WFA request_t) / AIM_FLAG_RES8 ({idx, score}) give the same scores as the
    default 16-B / 24-B structs on every kernel; RES8 + BACKTRACE is rejected."""
    from aim_amd import capi, engine
    ms, rs = engine.launcher_sizes(algo, l, err)
    n = 3000 if l <= 100 else (300 if l <= 1000 else 40)
    req, pat, txt = engine.gen_pairs(4242, 1000, n, l, err, rs)
    base, _ = engine.align(engine.make_params(algo, ms, rs, **kw), req, pat, txt)
    res, _ = engine.align(engine.make_params(algo, ms, rs, req8=req8, res8=res8, **kw), req, pat, txt)
    assert np.array_equal(res["score"], base["score"]) and np.array_equal(res["idx"], req["idx"])
    if not res8:
        for f in ("max_operations", "begin_offset", "end_offset", "status"):
            assert np.array_equal(res[f], base[f])
    with pytest.raises(capi.AimError) as e:
        engine.align(engine.make_params(algo, ms, rs, res8=True, backtrace=True), req, pat, txt)
    assert e.value.code == capi.AIM_EINVAL
    if req8:   # CIGAR output with the 8-B request
        b2, o2 = engine.align(engine.make_params(algo, ms, rs, backtrace=True, **kw), req, pat, txt, check=False)
        r2, p2 = engine.align(engine.make_params(algo, ms, rs, backtrace=True, req8=True, **kw), req, pat, txt, check=False)
        assert engine.format_output(r2, p2, True) == engine.format_output(b2, o2, True) if (b2["status"] == 0).all() else True


# ------------------------------------------------------------------ round 2: packed input, compact CIGAR, slots
PIPE_CASES = [("wfa", 100, 0.01, 5000, dict(reduce=True)), ("wfa", 100, 0.01, 5000, dict(reduce=True, backtrace=True)),
              ("wfa", 100, 0.05, 3000, dict(reduce=True, backtrace=True)), ("wfa", 1000, 0.05, 200, dict(reduce=True, backtrace=True)),
              ("nw", 100, 0.05, 3000, dict(backtrace=True)), ("swg", 100, 0.02, 3000, dict(backtrace=True)),
              ("nw", 700, 0.10, 60, dict(backtrace=True)), ("swg", 150, 0.03, 1000, dict())]


@pytest.mark.parametrize("algo,l,err,n,kw", PIPE_CASES)
def test_packed_input_and_compact_cigar_match_default_path(gpu, algo, l, err, n, kw):
    """aim_set_submit / aim_set_wait with a PACKED batch (2 bits per base + raw side list for pairs with non-ACGT bytes) and
    device-side CIGAR run-length encoding give byte-identical output text to the default push / launch / pull path."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes(algo, l, err)
    req, pat, txt = engine.gen_pairs(2025, 7, n, l, err, rs)
    for i in range(0, n, 9):                      # pairs that cannot be packed: 'N' matches 'N', 'n' does not match 'N'
        pat[i, i % (l // 2)] = ord("N")
    for i in range(4, n, 31):
        j = (3 * i) % (l // 2)
        pat[i, j] = ord("N"); txt[i, j] = ord("N")
    bt = kw.get("backtrace", False)
    params = engine.make_params(algo, ms, rs, **kw)
    base_res, base_ops = engine.align(params, req, pat, txt, check=False)
    want = engine.format_output(base_res, base_ops, bt) if (base_res["status"] == 0).all() else None
    packed = engine.pack_batch(req, pat, txt)
    assert len(packed[2]) >= n // 9
    with engine.DeviceSet(1) as s:
        s.configure_slots(params, n, slots=2, max_raw=n, max_runs=(n * 2 * rs if bt else 0))
        # (a) packed in, default results out
        s.submit(0, 0, req, packed=packed, want_ops=bt)
        out = s.wait(0, 0, check=False)
        for f in ("score", "status", "begin_offset", "end_offset", "idx"):
            assert np.array_equal(out["res"][f], base_res[f]), f
        if want is not None:
            assert engine.format_output(out["res"], out.get("ops"), bt) == want
        if bt:
            # (b) ASCII in, compact CIGAR out; (c) packed in, compact CIGAR out -- on the other slot
            for slot, pk in ((1, None), (0, packed)):
                if pk is None:
                    s.submit(0, slot, req, pat, txt, cigar_runs_cap=n * 2 * rs)
                else:
                    s.submit(0, slot, req, packed=pk, cigar_runs_cap=n * 2 * rs)
                out = s.wait(0, slot, check=False)
                assert np.array_equal(out["cig"]["score"], base_res["score"]) and np.array_equal(out["cig"]["idx"], base_res["idx"])
                assert np.array_equal(out["cig"]["status"], base_res["status"].astype(np.uint16))
                if want is not None:
                    assert engine.format_output_runs(out["cig"], out["runs"]) == want
                if pk is not None:   # fused kernels: slotted run buffer (lane) / void runs of the side list's pairs left behind as holes (group)
                    assert len(out["runs"]) >= int(out["cig"]["n_runs"].sum())
                else:
                    assert int(out["cig"]["n_runs"].sum()) == len(out["runs"])
            # (d) ADVICE r03: packed in (side list not empty), compact CIGAR AND result_t + ops rows out of the same submit -- the raw side
            # pass must re-align the side list's pairs into BOTH outputs
            s.submit(0, 1, req, packed=packed, cigar_runs_cap=n * 2 * rs, want_ops=True)
            out = s.wait(0, 1, check=False)
            for f in ("score", "status", "begin_offset", "end_offset", "idx"):
                assert np.array_equal(out["res"][f], base_res[f]), f
            assert np.array_equal(out["cig"]["score"], base_res["score"]) and np.array_equal(out["cig"]["status"], base_res["status"].astype(np.uint16))
            if want is not None:
                assert engine.format_output(out["res"], out["ops"], True) == want
                assert engine.format_output_runs(out["cig"], out["runs"]) == want


def test_compact_cigar_bytes_per_pair_and_overflow(gpu):
    """l=100 e=1 % with CIGAR: the compact form is <= 32 B per pair (header 16 B + ~3 runs), against 24 + 224 B for the
    default structs + ops rows; a run buffer that is too small is reported (AIM_ENOMEM), never silently truncated."""
    from aim_amd import capi, engine
    ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
    n = 1 << 16
    req, pat, txt = engine.gen_pairs(1, 0, n, 100, 0.01, rs)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=True, req8=True)
    with engine.DeviceSet(1) as s:
        s.configure_slots(params, n, slots=1, max_raw=0, max_runs=8 * n)
        s.submit(0, 0, req, pat, txt, cigar_runs_cap=8 * n)
        out = s.wait(0, 0)
        per_pair = (out["cig"].nbytes + out["runs"].nbytes) / n
        assert per_pair <= 32.0, per_pair
        base_res, base_ops = engine.align(engine.make_params("wfa", ms, rs, reduce=True, backtrace=True), req, pat, txt)
        assert engine.format_output_runs(out["cig"], out["runs"]) == engine.format_output(base_res, base_ops, True)
        s.submit(0, 0, req, pat, txt, cigar_runs_cap=n)          # one run per pair is not enough
        with pytest.raises(capi.AimError) as e:
            s.wait(0, 0)
        assert e.value.code == capi.AIM_ENOMEM


def test_large_batch_host_scans_report_the_offending_pair(gpu):
    """aim_set_submit's length check and aim_set_wait's status scan run on several host threads from 2^19 pairs up (they sat on the
    caller's critical path at 4 M pairs per batch): a bad length / a failing pair anywhere in such a batch is still reported, by index."""
    from aim_amd import capi, engine
    n, rs, ms = (1 << 19) + 4097, 32, 3
    req, pat, txt = engine.gen_pairs(5, 0, n, 24, 0.04, rs)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=True, req8=True)
    with engine.DeviceSet(1) as s:
        s.configure_slots(params, n, slots=1, max_raw=0, max_runs=8 * n)
        bad = req.copy()
        bad["pattern_len"][n - 5] = rs + 1                      # in the last thread's share
        with pytest.raises(capi.AimError) as e:
            s.submit(0, 0, bad, pat, txt, cigar_runs_cap=8 * n)
        assert e.value.code == capi.AIM_EINVAL and ("pair %d" % (n - 5)) in str(e.value)
        s._inflight.pop((0, 0), None)
        s.submit(0, 0, req, pat, txt, cigar_runs_cap=8 * n)     # the same batch with honest lengths goes through
        out = s.wait(0, 0)
        base_res, base_ops = engine.align(engine.make_params("wfa", ms, rs, reduce=True, backtrace=True), req[:4096], pat[:4096], txt[:4096])
        assert np.array_equal(out["cig"]["score"][:4096], base_res["score"])
        s.submit(0, 0, req, pat, txt, cigar_runs_cap=n)            # one run per pair is not enough: the pairs the buffer has no room for are flagged
        with pytest.raises(capi.AimError) as e:
            s.wait(0, 0)
        assert e.value.code == capi.AIM_ENOMEM


def test_two_slots_pipeline_many_batches(gpu):
    """Double buffering: batches alternate between two slots of one device, each slot's results land in its own buffers;
    the concatenation equals one big default launch. Also: a slot refuses a second submit before its wait."""
    from aim_amd import capi, engine
    ms, rs = engine.launcher_sizes("wfa", 100, 0.02)
    nb, per = 12, 4096
    req, pat, txt = engine.gen_pairs(314, 0, nb * per, 100, 0.02, rs)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=True)
    base_res, base_ops = engine.align(params, req, pat, txt)
    want = engine.format_output(base_res, base_ops, True)
    got = []
    with engine.DeviceSet(1) as s:
        s.configure_slots(params, per, slots=2, max_raw=per, max_runs=16 * per)
        def sub(b):
            lo, hi = b * per, (b + 1) * per
            s.submit(0, b & 1, req[lo:hi], packed=engine.pack_batch(req[lo:hi], pat[lo:hi], txt[lo:hi]), cigar_runs_cap=16 * per)
        sub(0)
        with pytest.raises(capi.AimError) as e:
            sub(2)                                   # slot 0 is busy
        assert e.value.code == capi.AIM_ESTATE
        for b in range(nb):
            if b + 1 < nb:
                sub(b + 1)
            out = s.wait(0, b & 1)
            got.append(engine.format_output_runs(out["cig"], out["runs"]))
    assert b"".join(got) == want


@pytest.mark.parametrize("rs,l", [(112, 100), (80, 70)])
@pytest.mark.parametrize("reduce", [True, False])
def test_wfa_lane_dynamic_bounds_shape(gpu, rs, l, reduce):
    """MAX_SCORE 6..10 at the default penalties runs on wfa_lane_kernel's dynamic-bounds instantiation (score-only): wavefronts
    reach 11 and 13 diagonals, so WFA-adaptive's reduction (wfa.c:69-140) fires and klo/khi/null flags become per-pair data.
    Error rates up to 10 % make the reduction actually cut, pairs exceed the cap, and non-ACGT pairs take the raw path."""
    from aim_amd import capi, engine
    import ctypes as C
    lib = capi.load()
    for ms in (6, 8, 10):
        for err in (0.02, 0.05, 0.10):
            params = engine.make_params("wfa", ms, rs, reduce=reduce)
            assert lib.aim_kernel_name(C.byref(params)) == b"wfa_lane_kernel"
            req, pat, txt = engine.gen_pairs(900 + ms, 0, 6000, l, err, rs)
            for i in range(0, 6000, 97):
                pat[i, i % (l // 2)] = ord("N")
            res, _, ores = _compare("wfa", params, req, pat, txt)
            assert (res["score"] == ms + 1).any() or err < 0.05      # the cap is exercised
    # For MAX_SCORE <= 10 at (3,4,1) the reduction can change klo/khi but provably never a SCORE: the first wavefront of >= 10
    # diagonals is score 9's; a cut there can only reach the end test of score 10 through M[10][ak], whose five sources are
    # M[7][ak], M[5][ak-1], M[5][ak+1], I[9][ak-1], D[9][ak+1] -- and diagonals ak-1 .. ak+1 are never cut (top_limit / bottom_limit,
    # wfa.c:111-126). So with and without -r the scores must be EQUAL here (the judge's reference runs agree: same md5); pairs on
    # which the reduction does change the score need MAX_SCORE >= 21 and are tested below on the kernels that take that shape.
    req, pat, txt = engine.gen_pairs(4321, 0, 20000, l, 0.10, rs)
    a, _ = engine.align(engine.make_params("wfa", 10, rs, reduce=True), req, pat, txt)
    b, _ = engine.align(engine.make_params("wfa", 10, rs, reduce=False), req, pat, txt)
    assert np.array_equal(a["score"], b["score"])


@pytest.mark.parametrize("env", [dict(), dict(AIM_FORCE_WAVE="1"), dict(AIM_GROUP_G="64"), dict(AIM_GROUP_G="4")])
@pytest.mark.parametrize("bt", [False, True])
def test_reduction_changes_scores_on_constructed_pairs(gpu, monkeypatch, env, bt):
    """Constructed pairs on which WFA-adaptive's reduction cuts the diagonal the optimum needs (reduce_changes_score.json): the
    HIP kernels must reproduce BOTH scores (21 without -r; 22 / 25 with) and the oracle's CIGARs -- i.e. the reduction is shown to
    act, not merely not to break anything. Replicated to fill several wavefronts."""
    from aim_amd import engine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d, req, pat, txt, plain, red = reduce_cases()
    reps = 67
    req = np.tile(req, reps); pat = np.tile(pat, (reps, 1)); txt = np.tile(txt, (reps, 1))
    req["idx"] = np.arange(len(req))
    for reduce, want in ((False, plain), (True, red)):
        params = engine.make_params("wfa", d["max_score"], d["read_size"], reduce=reduce, backtrace=bt)
        res, _, _ = _compare("wfa", params, req, pat, txt)
        assert np.array_equal(res["score"], np.tile(want, reps))


def test_host_cli_falls_back_to_ops_rows_when_the_run_buffer_overflows(gpu, tmp_path):
    """Found by tools/fuzz_cli.py --focus dplong in round 2: SWG with MAX_SCORE as '+infinity' (S2) on l = 700 reads produces
    alignments with hundreds of one-operation runs, more than the READ_SIZE/4 + 2 per pair the compact-CIGAR buffer holds.
    The host must then gather result_t + ops rows for that batch (like the reference) instead of giving up."""
    import subprocess, sys
    from conftest import ROOT
    from aim_amd import engine
    l, err, cost = 700, 0.02, dict(mismatch=3, gap_o=3, gap_e=2)
    ms, rs = engine.launcher_sizes("swg", l, err, **cost)
    req, pat, txt = engine.gen_pairs(1234, 0, 40, l, err, rs)
    inp = tmp_path / "in.seq"
    inp.write_bytes(engine.pairs_to_text(req, pat, txt))
    common = ["-i", str(inp), "-l", str(l), "-e", str(err), "-n", "2", "-d", "1", "-b", "-x", "3", "-g", "3", "-a", "2"]
    rh = subprocess.run([sys.executable, "-m", "aim_amd.launch", "swg", "-o", str(tmp_path / "h.out")] + common, capture_output=True, text=True,
                        cwd=tmp_path, env=dict(os.environ, PYTHONPATH=ROOT))
    ro = subprocess.run([os.path.join(ROOT, "oracle", "oracle_cli"), "swg", "-o", str(tmp_path / "o.out")] + common, capture_output=True, text=True,
                        cwd=tmp_path)
    assert (rh.returncode == 0) == (ro.returncode == 0), rh.stdout + rh.stderr
    if ro.returncode == 0:
        assert (tmp_path / "h.out").read_bytes() == (tmp_path / "o.out").read_bytes()
        runs = (tmp_path / "o.out").read_bytes().split(b"\n")[1]
        assert sum(c in b"MXID" for c in runs) > rs // 4 + 2      # the case really overflows the compact buffer


@pytest.mark.parametrize("env,expect_fallback", [(dict(), False), (dict(AIM_GROUP_WLDS="64"), True), (dict(AIM_GROUP_WLDS="0"), False),
                                                 (dict(AIM_GROUP_G="16"), False), (dict(AIM_GROUP_G="64", AIM_GROUP_WLDS="64"), True)])
def test_wfa_group_narrow_window_and_two_pairs_per_wavefront(gpu, monkeypatch, env, expect_fallback):
    """Round 2: for MAX_SCORE >= 127 with WFA-adaptive the ring rows in LDS are 128 entries addressed modulo 128 instead of
    2*MAX_SCORE+3 homes, and 32 lanes own a pair (two pairs per wavefront). A wavefront wider than the row sends its pair to
    the general kernel (forced here with 64-entry rows); every combination stays bit-exact, with and without CIGAR."""
    from aim_amd import engine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ms, rs = engine.launcher_sizes("wfa", 1000, 0.05)
    req, pat, txt = engine.gen_pairs(2718, 0, 3001, 1000, 0.05, rs)
    pat[7, 100] = ord("N")
    for kw in (dict(reduce=True), dict(reduce=True, backtrace=True)):
        params = engine.make_params("wfa", ms, rs, **kw)
        with engine.DeviceSet(1) as ds:
            ds.align(params, req, pat, txt)
            fb = ds.fallback_pairs(0)
            plan = ds.plan_describe(0)
        assert "wfa_group_kernel" in plan
        assert (fb > 1) == expect_fallback, (fb, plan)
        _compare("wfa", params, req, pat, txt)
    if not env:
        assert "G=16" in plan      # narrow rows: four pairs per wavefront (round 3)


def test_slots_on_several_devices_and_empty_batches(gpu):
    """aim_set_submit / aim_set_wait on every (device, slot) of a set (the same physical GPU three times here), an empty
    batch, and aim_pack_batch's side-list overflow report."""
    import ctypes as C
    from aim_amd import capi, engine
    lib = capi.load()
    ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
    n = 5000
    req, pat, txt = engine.gen_pairs(99, 0, 3 * n, 100, 0.01, rs)
    params = engine.make_params("wfa", ms, rs, reduce=True, req8=True, res8=True)
    base, _ = engine.align(engine.make_params("wfa", ms, rs, reduce=True), req, pat, txt)
    with engine.DeviceSet(device_ids=[0, 0, 0]) as s:
        s.configure_slots(params, n, slots=2, max_raw=16, max_runs=0)
        for d in range(3):
            lo = d * n
            s.submit(d, d & 1, req[lo:lo + n], packed=engine.pack_batch(req[lo:lo + n], pat[lo:lo + n], txt[lo:lo + n]))
        for d in (2, 0, 1):                                  # waits in any order
            out = s.wait(d, d & 1)
            assert np.array_equal(out["res"]["score"], base["score"][d * n:(d + 1) * n])
            assert np.array_equal(out["res"]["idx"], req["idx"][d * n:(d + 1) * n])
        s.submit(1, 0, req[:0], pat[:0], txt[:0])            # empty batch
        assert len(s.wait(1, 0)["res"]) == 0
        with pytest.raises(capi.AimError) as e:
            s.wait(1, 0)                                     # nothing in flight
        assert e.value.code == capi.AIM_ESTATE
    dirty = pat.copy()
    dirty[:200, 3] = ord("N")
    pp = np.zeros((3 * n, engine.packed_row_dwords(rs)), dtype=np.uint32); pt = pp.copy()
    raw = np.zeros(16, dtype=np.uint32); rawp = np.zeros((16, rs), dtype=np.uint8); rawt = rawp.copy()
    nr = C.c_uint32()
    rc = lib.aim_pack_batch(C.byref(params), 3 * n, capi.ptr(engine.to_request8(req)), capi.ptr(dirty), capi.ptr(txt), capi.ptr(pp), capi.ptr(pt),
                            capi.ptr(raw), capi.ptr(rawp), capi.ptr(rawt), 16, C.byref(nr), 4)
    assert rc == capi.AIM_ENOMEM and nr.value == 200


def test_host_cli_pipeline_over_two_set_members(gpu, sample_bytes, ref_digests, tmp_path):
    """The host's job ring over gpus x slots (the same physical GPU entered twice, small batches so that every (device, slot)
    is used several times): output identical to the reference digests, with packed input and with --no-pack / --full-ops."""
    import subprocess
    from conftest import ROOT
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    inp = tmp_path / "sample"
    inp.write_bytes(sample_bytes)
    for extra, key in ((["--backtrace", "--reduce"], "wfa_reduce_backtrace"), (["--reduce"], "wfa_score_only"),
                       (["--backtrace", "--reduce", "--no-pack", "--full-ops"], "wfa_reduce_backtrace")):
        out = tmp_path / "out"
        r = subprocess.run([host, str(inp), str(out), "20000", "--algo", "wfa", "--max-score", "5", "--read-size", "112", "--device-ids", "0,0",
                            "--slots", "2", "--batch", "1500"] + extra, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "14 batch(es)" in r.stdout and "2 device(s) x 2 slot(s)" in r.stdout, r.stdout
        assert md5(out.read_bytes()) == ref_digests[key]


# ------------------------------------------------------------------ round 3: one kernel per batch (packed rows in, compact CIGAR out)
def _oracle_text(algo, params, req, pat, txt):
    from oracle import oracle
    op = _oracle_params(oracle, params, algo)
    ores, oops, _ = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=8)
    return ores, oracle.format_output(ores, oops, bool(op.backtrace))


def _fused(params, req, pat, txt, runs_cap=None, expect_kernel="wfa_lane_packed_kernel", slots=1):
    """One packed batch through aim_set_submit / aim_set_wait; returns (out, plan line)."""
    from aim_amd import capi, engine
    n = len(req)
    bt = bool(params.flags & capi.FLAG_BACKTRACE)
    packed = engine.pack_batch(req, pat, txt)
    if runs_cap is None:
        runs_cap = 8 * n + 64
    with engine.DeviceSet(1) as s:
        s.configure_slots(params, max(n, 1), slots=slots, max_raw=max(n, 1), max_runs=(runs_cap if bt else 0))
        s.submit(0, 0, req, packed=packed, cigar_runs_cap=(runs_cap if bt else 0))
        out = s.wait(0, 0, check=False)
        plan = s.plan_describe(0)
    if expect_kernel:
        assert plan.startswith(expect_kernel), plan
    return out, plan


FUSED_SHAPES = [(100, 0.01, 112, 5), (70, 0.01, 80, 4), (100, 0.01, 112, 3), (150, 0.005, 160, 4), (140, 0.005, 144, 5), (170, 0.005, 176, 5)]


@pytest.mark.parametrize("l,err,rs,ms", FUSED_SHAPES)
@pytest.mark.parametrize("bt", [False, True])
def test_fused_packed_lane_kernel_against_oracle(gpu, l, err, rs, ms, bt):
    """SURVEY 8f-1 / 8f-2, VERDICT r02 item 1: a packed batch whose configuration wfa_lane_packed_kernel takes is ONE kernel --
    it reads the 2-bit rows itself and, with BACKTRACE, writes aim_cigar_t + runs itself. Compared with the ORACLE (scores,
    output text) -- not with the default path -- on every READ_SIZE the kernel is built for, incl. pairs that exceed the cap,
    pairs with bytes outside A/C/G/T (raw side pass) and a ragged batch tail."""
    from aim_amd import engine
    n = 3000 + 37
    req, pat, txt = engine.gen_pairs(77 + l, 3, n, l, err, rs)
    r2, p2, t2 = engine.gen_pairs(78 + l, 3, 500, l, min(0.04, (rs - l - 0.5) / l), rs)   # some pairs beyond MAX_SCORE
    req[1000:1500], pat[1000:1500], txt[1000:1500] = r2, p2, t2
    req["idx"] = np.arange(n)                                          # the oracle numbers its output 0 .. n-1
    for i in range(0, n, 11):                                          # pairs that cannot be packed
        pat[i, i % (l // 2)] = ord("N")
    for i in range(5, n, 53):
        j = (3 * i) % (l // 2)
        pat[i, j] = ord("N"); txt[i, j] = ord("N")
    params = engine.make_params("wfa", ms, rs, backtrace=bt, reduce=True, req8=True, res8=not bt)
    ores, want = _oracle_text("wfa", params, req, pat, txt)
    assert (ores["score"] == ms + 1).any() and (ores["score"] <= ms).any()
    out, _ = _fused(params, req, pat, txt)
    if bt:
        assert np.array_equal(out["cig"]["score"], ores["score"]) and np.array_equal(out["cig"]["idx"], req["idx"])
        assert (out["cig"]["status"] == 0).all()
        assert engine.format_output_runs(out["cig"], out["runs"]) == want
    else:
        assert np.array_equal(out["res"]["score"], ores["score"]) and np.array_equal(out["res"]["idx"], req["idx"])


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129, 1000])
def test_fused_packed_lane_kernel_ragged_and_edge_pairs(gpu, n):
    from aim_amd import engine
    req, pat, txt = engine.gen_pairs(5, 0, n, 100, 0.01, 112)
    for ms in (0, 2, 5):
        for bt in (False, True):
            params = engine.make_params("wfa", ms, 112, backtrace=bt, reduce=True)
            ores, want = _oracle_text("wfa", params, req, pat, txt)
            out, _ = _fused(params, req, pat, txt)
            if bt:
                assert engine.format_output_runs(out["cig"], out["runs"]) == want
            else:
                assert np.array_equal(out["res"]["score"], ores["score"])
    if n == 65:   # the hand-made edge pairs (empty sequences, all-mismatch, lower case, N runs, 112-base rows)
        req, pat, txt = _mk(EDGE_PAIRS, 112)
        req["idx"] = np.arange(len(req))
        for bt in (False, True):
            params = engine.make_params("wfa", 5, 112, backtrace=bt)
            ores, want = _oracle_text("wfa", params, req, pat, txt)
            out, _ = _fused(params, req, pat, txt)
            if bt:
                assert engine.format_output_runs(out["cig"], out["runs"]) == want
            else:
                assert np.array_equal(out["res"]["score"], ores["score"])


def test_fused_packed_lane_kernel_run_lists_equal_the_rle_kernel(gpu, monkeypatch):
    """The run lists the fused kernel emits are the lists cigar_rle_kernel produces from the ops rows of the default kernel
    (same runs, same n_runs, same status) -- and a run buffer with fewer than 3 runs per pair switches the fused kernel from
    its slotted layout to bump allocation without changing them."""
    from aim_amd import engine
    n = 20000
    req, pat, txt = engine.gen_pairs(91, 0, n, 100, 0.01, 112)
    r2, p2, t2 = engine.gen_pairs(92, 0, 4000, 100, 0.03, 112)
    req[3000:7000], pat[3000:7000], txt[3000:7000] = r2, p2, t2
    req["idx"] = np.arange(n)
    params = engine.make_params("wfa", 5, 112, backtrace=True, reduce=True)

    def lists(out):
        c, r = out["cig"], out["runs"]
        return [tuple(r[int(c["run_offset"][i]): int(c["run_offset"][i]) + int(c["n_runs"][i])]) for i in range(len(c))]
    fused, _ = _fused(params, req, pat, txt)
    bump, _ = _fused(params, req, pat, txt, runs_cap=3 * n - 1000)          # < 3 n: no slots (the batch has ~2.5 n runs)
    monkeypatch.setenv("AIM_NO_LANE_PK", "1")
    unfused, plan = _fused(params, req, pat, txt, expect_kernel="wfa_lane_kernel")
    a, b, c = lists(fused), lists(bump), lists(unfused)
    assert a == c and b == c
    assert np.array_equal(fused["cig"]["status"], unfused["cig"]["status"]) and np.array_equal(fused["cig"]["score"], unfused["cig"]["score"])
    assert len(bump["runs"]) == int(bump["cig"]["n_runs"].sum())             # bump allocation leaves no holes


def test_fused_packed_dynamic_bounds_shape_against_oracle(gpu):
    """MAX_SCORE 6..10 score-only on packed batches: the dynamic-bounds score loop inside wfa_lane_packed_kernel, READ_SIZE 80 .. 176."""
    from aim_amd import engine
    for l, rs in ((100, 112), (70, 80), (150, 160), (130, 144), (165, 176)):
        for ms, err in ((6, 0.02), (8, 0.05), (10, 0.10), (10, 0.02)):
            if l + int(np.ceil(l * err)) + 1 > rs:
                continue
            n = 4000
            req, pat, txt = engine.gen_pairs(900 + ms + l, 0, n, l, err, rs)
            for i in range(0, n, 97):
                pat[i, i % (l // 2)] = ord("N")
            for reduce in (True, False):
                params = engine.make_params("wfa", ms, rs, reduce=reduce, req8=True, res8=True)
                ores, _ = _oracle_text("wfa", params, req, pat, txt)
                out, _ = _fused(params, req, pat, txt)
                assert np.array_equal(out["res"]["score"], ores["score"]), (l, rs, ms, err, reduce)


GROUP_FUSED = [(100, 0.05, 3000, dict()), (100, 0.10, 2000, dict()), (150, 0.02, 3000, dict()), (250, 0.05, 1000, dict()),
               (1000, 0.05, 300, dict()), (100, 0.05, 2000, dict(mismatch=4, gap_o=6, gap_e=2))]


@pytest.mark.parametrize("l,err,n,cost", GROUP_FUSED)
@pytest.mark.parametrize("bt", [False, True])
def test_fused_group_kernel_against_oracle(gpu, l, err, n, cost, bt):
    """wfa_group_kernel on a packed batch (it copies the 2-bit rows into its LDS image itself) and, with BACKTRACE, the compact
    CIGAR straight from wfa_group_tb_kernel -- against the oracle, incl. non-ACGT pairs (raw side pass)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("wfa", l, err, **cost)
    req, pat, txt = engine.gen_pairs(31 + l, 0, n, l, err, rs)
    for i in range(0, n, 13):
        pat[i, i % (l // 2)] = ord("N")
    for reduce in (True, False):
        params = engine.make_params("wfa", ms, rs, backtrace=bt, reduce=reduce, req8=True, res8=not bt, **cost)
        ores, want = _oracle_text("wfa", params, req, pat, txt)
        out, plan = _fused(params, req, pat, txt, runs_cap=n * (2 * ms + 8), expect_kernel="wfa_group_kernel")
        assert "packed_in=1" in plan and ("runs_out=%d" % int(bt)) in plan, plan
        if bt:
            assert np.array_equal(out["cig"]["score"], ores["score"]) and (out["cig"]["status"] == 0).all()
            assert engine.format_output_runs(out["cig"], out["runs"]) == want
            assert len(out["runs"]) >= int(out["cig"]["n_runs"].sum())    # (holes: the void runs the packed kernel wrote for the side list's pairs)
        else:
            assert np.array_equal(out["res"]["score"], ores["score"])


@pytest.mark.parametrize("env", [dict(AIM_GROUP_WLDS="64"), dict(AIM_SCRATCH_GB="4"), dict(AIM_GROUP_G="64"), dict(AIM_GROUP_G="2")])
def test_fused_group_kernel_todo_list_chunks_and_plans(gpu, monkeypatch, env):
    """The side roads of the fused group path: a narrow LDS window that pairs outgrow (they reach the general kernel through the
    to-do list: their packed rows are expanded for it and its ops rows are run-length encoded afterwards), a scratch bound that
    splits the batch into several compute + traceback launches, and forced lanes-per-pair plans -- default ABI and fused I/O."""
    from aim_amd import engine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    l, err, n = 1000, 0.05, (12000 if "AIM_SCRATCH_GB" in env else 1500)   # 12 000 history regions of 193 KB (252 rows of 96 cells) do not fit half of a 4 GB bound: three chunks
    ms, rs = engine.launcher_sizes("wfa", l, err)
    req, pat, txt = engine.gen_pairs(17, 0, n, l, err, rs)
    pat[5, 17] = ord("N")
    for bt in (True, False):
        params = engine.make_params("wfa", ms, rs, backtrace=bt, reduce=True)
        ores, want = _oracle_text("wfa", params, req, pat, txt)
        out, plan = _fused(params, req, pat, txt, runs_cap=n * (2 * ms + 8), expect_kernel="wfa_group_kernel")
        if "AIM_SCRATCH_GB" in env and bt:
            chunk = int(plan.split("chunk=")[1].split()[0])
            assert chunk < n, plan                                          # the batch really ran in several chunks
        if bt:
            assert engine.format_output_runs(out["cig"], out["runs"]) == want
        else:
            assert np.array_equal(out["res"]["score"], ores["score"])
        _compare("wfa", params, req, pat, txt)                              # default ABI (ASCII rows in, result_t + ops rows out)


LANE_SHAPES = [(150, 0.01), (150, 0.02), (125, 0.01), (100, 0.02), (70, 0.03), (160, 0.01), (100, 0.01)]


@pytest.mark.parametrize("l,err", LANE_SHAPES)
@pytest.mark.parametrize("bt", [False, True])
def test_lane_shapes_through_the_default_abi(gpu, l, err, bt):
    """the one-pair-per-lane kernels on the shapes real short reads have -- READ_SIZE 136 .. 176 (l = 125 .. 160)
    and CIGAR at MAX_SCORE 6..10 -- through the DEFAULT ABI (ASCII rows in, result_t + ops rows out): the rows are packed on the
    device, wfa_lane_packed_kernel aligns them (dynamic-bounds score loop + LDS history for the walk at MAX_SCORE > 5), pairs with
    non-ACGT bytes reach the general kernel through the to-do list. Bit-exact against the oracle incl. begin_offset and ops."""
    from aim_amd import capi, engine
    import ctypes as C
    ms, rs = engine.launcher_sizes("wfa", l, err)
    if ms > 10:
        ms = 10          # run-time cap inside the shape (pairs beyond it report MAX_SCORE + 1)
    n = 5000 + 13
    req, pat, txt = engine.gen_pairs(400 + l, 0, n, l, err, rs)
    for i in range(0, n, 101):
        pat[i, i % (l // 2)] = ord("N")
    for reduce in (True, False):
        params = engine.make_params("wfa", ms, rs, backtrace=bt, reduce=reduce)
        want = b"wfa_lane_kernel" if (rs in (80, 112) and (ms <= 5 or not bt)) else b"wfa_lane_packed_kernel"
        assert capi.load().aim_kernel_name(C.byref(params)) == want
        res, _, ores = _compare("wfa", params, req, pat, txt)
        assert (ores["score"] <= ms).any()
    with engine.DeviceSet(1) as s:       # the non-ACGT pairs went through the to-do list
        s.configure(params, n)
        s.push(0, req, pat, txt)
        s.launch()
        if want == b"wfa_lane_packed_kernel":
            assert s.fallback_pairs(0) == len(range(0, n, 101))


@pytest.mark.parametrize("l,err,rs", [(100, 0.02, 112), (100, 0.05, 112), (70, 0.03, 80), (150, 0.02, 160), (165, 0.03, 176)])
def test_fused_packed_dynamic_bounds_shape_with_cigar(gpu, l, err, rs):
    """MAX_SCORE 6..10 WITH the compact CIGAR on packed batches: wfa_scores_dynamic<HIST> + wfa_backtrace_dynamic inside
    wfa_lane_packed_kernel, against the oracle's output text (reduction on and off, pairs beyond the cap, non-ACGT pairs)."""
    from aim_amd import engine
    n = 4000 + 9
    req, pat, txt = engine.gen_pairs(1300 + l, 0, n, l, err, rs)
    for i in range(0, n, 89):
        pat[i, i % (l // 2)] = ord("N")
    for ms in (6, 8, 10):
        for reduce in (True, False):
            params = engine.make_params("wfa", ms, rs, backtrace=True, reduce=reduce, req8=True)
            ores, want = _oracle_text("wfa", params, req, pat, txt)
            out, _ = _fused(params, req, pat, txt, runs_cap=24 * n)
            assert np.array_equal(out["cig"]["score"], ores["score"]) and (out["cig"]["status"] == 0).all()
            assert engine.format_output_runs(out["cig"], out["runs"]) == want, (l, ms, reduce)


def test_host_cli_packed_input_file(gpu, sample_bytes, err_bytes, ref_digests, tmp_path):
    """`host --packed-input` reads the packed batch file `host --pack-only` writes -- same output file as the text
    input, byte for byte (reference digests on the sample set; real reads with N through the raw side list; the reference's
    partition rule ending inside a batch; a file packed for another READ_SIZE is refused like an over-length read)."""
    import subprocess
    from conftest import ROOT
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    (tmp_path / "s.seq").write_bytes(sample_bytes)
    (tmp_path / "e.seq").write_bytes(err_bytes)
    base = ["--algo", "wfa", "--max-score", "5", "--read-size", "112", "--reduce"]
    for name, n in (("s", 20000), ("e", 2000)):
        r = subprocess.run([host, str(tmp_path / (name + ".seq")), str(tmp_path / "x"), str(n)] + base + ["--pack-only", str(tmp_path / (name + ".aimpk"))],
                           capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
    for extra, key in ((["--backtrace"], "wfa_reduce_backtrace"), ([], "wfa_score_only")):
        r = subprocess.run([host, str(tmp_path / "s.aimpk"), str(tmp_path / "p.out"), "20000", "--packed-input"] + base + extra, capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0 and "packed batch file" in r.stdout, r.stdout + r.stderr
        assert md5((tmp_path / "p.out").read_bytes()) == ref_digests[key]
        for n_arg, nd in (("2000", "1"), ("1001", "4")):            # the partition rule: ROUND_UP_8(n / d) * d pairs, possibly ending inside the file's batch
            outs = []
            for inp, flag in (("e.seq", []), ("e.aimpk", ["--packed-input"])):
                r = subprocess.run([host, str(tmp_path / inp), str(tmp_path / "q.out"), n_arg, "--nr-dpus", nd] + flag + base + extra, capture_output=True, text=True, cwd=tmp_path)
                assert r.returncode == 0, r.stdout + r.stderr
                outs.append((tmp_path / "q.out").read_bytes())
            assert outs[0] == outs[1] and outs[0].count(b"\n") == (1 + bool(extra)) * min(2000, (int(n_arg) // int(nd) + 7) // 8 * 8 * int(nd))
    r = subprocess.run([host, str(tmp_path / "s.aimpk"), str(tmp_path / "p.out"), "20000", "--packed-input", "--read-size", "120"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and "READ LENGTH less than length of the input reads" in r.stdout and (tmp_path / "p.out").read_bytes() == b""
    r = subprocess.run([host, str(tmp_path / "s.seq"), str(tmp_path / "p.out"), "20000", "--packed-input"] + base, capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "not a packed batch file" in r.stderr


@pytest.mark.parametrize("env", [dict(AIM_GROUP_OVERLAP="1"), dict()])
def test_group_kernel_chunks_overlap_traceback_and_compute(gpu, monkeypatch, env):
    """AIM_GROUP_OVERLAP=1: a batch larger than two rounds of wfa_group's persistent grid runs as several chunks, the traceback
    kernel of chunk c on a second stream while chunk c + 1 is computed, two buffers of history regions alternating (measured
    slower than one launch, so not the default -- but the same chunk loop is what a batch takes whose history regions exceed the
    scratch bound). Results must not depend on it: default ABI and fused I/O against the oracle, 150 000 pairs, three chunks."""
    from aim_amd import engine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    l, err, n = 100, 0.05, 150000
    ms, rs = engine.launcher_sizes("wfa", l, err)
    req, pat, txt = engine.gen_pairs(2718, 0, n, l, err, rs)
    for i in range(0, n, 997):
        pat[i, i % 50] = ord("N")
    params = engine.make_params("wfa", ms, rs, backtrace=True, reduce=True)
    ores, want = _oracle_text("wfa", params, req, pat, txt)
    out, plan = _fused(params, req, pat, txt, runs_cap=n * 24, expect_kernel="wfa_group_kernel")
    chunk = int(plan.split("chunk=")[1].split()[0])
    assert (chunk < n) == bool(env), plan
    assert np.array_equal(out["cig"]["score"], ores["score"]) and engine.format_output_runs(out["cig"], out["runs"]) == want
    res, ops = engine.align(params, req, pat, txt)
    assert engine.format_output(res, ops, True) == want


# ------------------------------------------------------------------ round 4
@pytest.mark.parametrize("l,err,n,kw", [(16300, 0.01, 6, dict(reduce=True, backtrace=True)), (20000, 0.005, 6, dict(reduce=True)),
                                        (32000, 0.002, 4, dict(reduce=True, backtrace=True)), (32000, 0.002, 4, dict())])
def test_wfa_reads_up_to_the_int16_length_limit(gpu, l, err, n, kw):
    """The reference's lengths and WFA offsets are int16 (WFA/DPU-WRAM/common/common.h:98-100, 174-175): READ_SIZE < 32 760 is admitted
    (rounds 1-3 stopped at 16 376 for no reason the kernels have)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("wfa", l, err)
    assert 16376 <= rs < 32760
    req, pat, txt = engine.gen_pairs(77, 0, n, l, err, rs)
    _compare("wfa", engine.make_params("wfa", ms, rs, **kw), req, pat, txt)
    with pytest.raises(Exception):
        engine.align(engine.make_params("wfa", ms, 32760, **kw), req[:1], np.zeros((1, 32760), np.uint8), np.zeros((1, 32760), np.uint8))


def test_host_cli_writes_to_a_pipe(gpu, sample_bytes, ref_digests, tmp_path):
    """like the reference's fopen(out, "w"), the output may be a pipe / FIFO (pwrite fails there with ESPIPE): the text goes
    out with sequential write() in batch order. Whole-file digest through `host in /dev/stdout N | cat > file`."""
    import subprocess
    inp = tmp_path / "sample"
    inp.write_bytes(sample_bytes)
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    for flags, key in ((["--backtrace", "--reduce"], "wfa_reduce_backtrace"), (["--reduce"], "wfa_score_only")):
        fifo = tmp_path / "fifo"
        if fifo.exists():
            fifo.unlink()
        os.mkfifo(fifo)
        out = tmp_path / "piped.out"
        reader = subprocess.Popen("cat %s > %s" % (fifo, out), shell=True)
        r = subprocess.run([host, str(inp), str(fifo), "20000", "--algo", "wfa", "--max-score", "5", "--read-size", "112", "--batch", "3000",
                            "--threads", "6"] + flags, capture_output=True, text=True, cwd=str(tmp_path))
        assert reader.wait(timeout=60) == 0
        assert r.returncode == 0, r.stdout + r.stderr
        assert md5(out.read_bytes()) == ref_digests[key]


def test_two_host_threads_drive_chunked_group_launches_on_one_device(gpu, monkeypatch):
    """ADVICE r03 / VERDICT r03 weak 11: the second stream + events of wfa_group's chunked CIGAR launches belong to the slot, so two sets
    driven by two host threads on the same device cannot wait on each other's events. Both threads' results equal the oracle's."""
    import threading
    from aim_amd import engine
    from oracle import oracle
    monkeypatch.setenv("AIM_GROUP_OVERLAP", "1")
    monkeypatch.setenv("AIM_SCRATCH_GB", "1")             # several chunks per batch
    ms, rs = engine.launcher_sizes("wfa", 250, 0.05)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=True)
    jobs, errs = [], []
    for t in range(2):
        req, pat, txt = engine.gen_pairs(900 + t, 0, 20000, 250, 0.05, rs)
        ores, oops, _ = oracle.align_batch(_oracle_params(oracle, params, "wfa"), req["pattern_len"], req["text_len"], pat, txt, nthreads=8)
        jobs.append((req, pat, txt, oracle.format_output(ores, oops, True)))
    def drive(t):
        try:
            req, pat, txt, want = jobs[t]
            with engine.DeviceSet(1) as s:
                for _ in range(6):
                    res, ops = s.align(params, req, pat, txt)
                    assert "chunk=" in s.plan_describe(0) and "wfa_group_kernel" in s.plan_describe(0)
                    assert engine.format_output(res, ops, True) == want
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            errs.append((t, repr(e)))
    th = [threading.Thread(target=drive, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs


@pytest.mark.parametrize("shards,extra", [(3, ["--batch", "3000"]), (2, ["--batch", "2500", "--device-ids", "0,0,0,0"]), (8, ["--batch", "4000", "--threads", "16"]),
                                          (4, ["--batch", "3000", "--no-pin", "--slots", "1"])])
def test_host_cli_out_shards_concatenate_to_the_single_file(gpu, sample_bytes, err_full_bytes, ref_digests, tmp_path, shards, extra):
    """`--out-shards K` runs K lanes (pack pool, device set, format pool, writer, output file each) over K contiguous
    runs of batches; `cat <out>.0*` is byte-identical to the single file -- text and packed input, score-only and CIGAR, more lanes than
    devices and more devices than lanes, a lane count that leaves some shards empty, real reads with 'N' (raw side list)."""
    import glob
    import subprocess
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    inp = tmp_path / "sample"
    inp.write_bytes(sample_bytes)
    pk = tmp_path / "sample.aimpk"
    r = subprocess.run([host, str(inp), "/dev/null", "20000", "--read-size", "112", "--batch", "3000", "--pack-only", str(pk)], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    def run(src, n, flags, packed=False):
        for f in glob.glob(str(tmp_path / "out*")):
            os.unlink(f)
        cmd = [host, str(src), str(tmp_path / "out"), str(n), "--algo", "wfa", "--max-score", "5", "--read-size", "112", "--out-shards", str(shards)] + flags
        cmd += (["--packed-input"] if packed else []) + extra          # (a packed file's batches are taken as the file holds them)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stdout + r.stderr
        files = sorted(glob.glob(str(tmp_path / "out.*")))
        assert len(files) == shards and not os.path.exists(tmp_path / "out")
        assert ("%d lane(s)" % shards) in r.stdout and r.stdout.count("Retrieve results") == 1 and r.stdout.count("Copying data to DPU") == 1
        return b"".join(open(f, "rb").read() for f in files)
    assert md5(run(inp, 20000, ["--backtrace", "--reduce"])) == ref_digests["wfa_reduce_backtrace"]
    assert md5(run(inp, 20000, ["--reduce"])) == ref_digests["wfa_score_only"]
    assert md5(run(pk, 20000, ["--backtrace", "--reduce"], packed=True)) == ref_digests["wfa_reduce_backtrace"]
    assert md5(run(pk, 20000, ["--reduce"], packed=True)) == ref_digests["wfa_score_only"]
    err = tmp_path / "err"
    err.write_bytes(err_full_bytes)
    want = {c["name"]: c["output_md5"] for c in judge_dataset_cases()}
    assert md5(run(err, 15000, ["--backtrace", "--reduce", "--nr-dpus", "4"])) == want["err240727_wfa_mram_bt_red"]
    assert md5(run(err, 15000, ["--reduce", "--nr-dpus", "4"])) == want["err240727_wfa_sc_red"]


@pytest.mark.parametrize("l,err,n", [(10000, 0.01, 48), (4000, 0.02, 96), (2000, 0.05, 128), (16000, 0.01, 16), (5000, 0.05, 32)])
def test_wfa_adaptive_long_reads_on_the_group_kernel(gpu, l, err, n):
    """WFA-adaptive beyond READ_SIZE 2048 / MAX_SCORE 400 (the launcher's shapes for l = 10 000 e = 1 %: MAX_SCORE 500,
    READ_SIZE 10 112) runs on wfa_group_kernel (packed image in LDS, rows of 96 / 128, history regions + traceback kernel) instead of one
    pair per wavefront on wfa_wave_kernel -- bit-exact against the oracle, score-only and with CIGAR, through ASCII and packed batches."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("wfa", l, err)
    req, pat, txt = engine.gen_pairs(4100 + l, 0, n, l, err, rs)
    for kw in (dict(reduce=True), dict(reduce=True, backtrace=True)):
        params = engine.make_params("wfa", ms, rs, **kw)
        _compare("wfa", params, req, pat, txt)
    with engine.DeviceSet(1) as s:
        s.configure(engine.make_params("wfa", ms, rs, reduce=True), n)
        s.push(0, req, pat, txt); s.launch(); s.pull(0)
        assert s.plan_describe(0).startswith("wfa_group_kernel"), s.plan_describe(0)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=True, req8=True)
    want = _oracle_text("wfa", params, req, pat, txt)[1]
    with engine.DeviceSet(1) as s:
        cap = n * (rs // 4 + 2)
        s.configure_slots(params, n, slots=1, max_raw=n, max_runs=cap)
        s.submit(0, 0, req, packed=engine.pack_batch(req, pat, txt), cigar_runs_cap=cap)
        out = s.wait(0, 0)
    assert engine.format_output_runs(out["cig"], out["runs"]) == want


@pytest.mark.parametrize("l,err", [(100, 0.01), (100, 0.05), (100, 0.10), (70, 0.02), (60, 0.10), (90, 0.03), (104, 0.0), (150, 0.01), (150, 0.05), (130, 0.03), (165, 0.02), (40, 0.05)])
@pytest.mark.parametrize("bt", [False, True])
def test_nw_row_in_registers_kernel(gpu, monkeypatch, l, err, bt):
    """nw_reg_kernel (dp_reg.hpp: the DP row in registers, right-aligned, two int16 cells per VGPR) against the oracle --
    all three length relations it takes (plen < / == / == tlen + 1: the aliased boundary cell of quirk N1), the pairs it hands to
    nw_lane_kernel (plen >= tlen + 2, short outliers), READ_SIZE 72 .. 112, penalties other than the default, and equality with the
    LDS-row kernel alone (AIM_NO_NW_REG=1)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("nw", l, max(err, 0.01))
    n = 6000
    req, pat, txt = engine.gen_pairs(7000 + l, 0, n, l, err, rs)
    # outliers: a few short pairs and a few long tails
    for i in range(0, n, 97):
        req["pattern_len"][i] = max(1, l // 3)
    for i in range(5, n, 131):
        req["text_len"][i] = max(1, int(req["text_len"][i]) - 7)
    for mism, gap in ((3, 4), (2, 5), (7, 3)):
        params = engine.make_params("nw", ms, rs, backtrace=bt, mismatch=mism, gap=gap)
        res, ops, _ = _compare("nw", params, req, pat, txt)
        with engine.DeviceSet(1) as s:
            s.configure(params, n)
            s.push(0, req, pat, txt); s.launch(); s.pull(0)
            if rs > 176:
                assert s.plan_describe(0).startswith("dp_group_kernel"), s.plan_describe(0)
                continue
            assert s.plan_describe(0).startswith("nw_reg_kernel"), s.plan_describe(0)
            fb = s.fallback_pairs(0)
            keep = 9                                                         # round 5: up to 8 tail cells in the last row stay in the kernel (with CIGAR: their direction bits in a register)
            tails = int((req["pattern_len"] > req["text_len"] + keep).sum())
            assert tails <= fb <= tails + n // 40, (fb, tails)               # the to-do list: tail pairs + the short outliers
    monkeypatch.setenv("AIM_NO_NW_REG", "1")
    params = engine.make_params("nw", ms, rs, backtrace=bt)
    res2, ops2 = engine.align(params, req, pat, txt)
    monkeypatch.delenv("AIM_NO_NW_REG")
    res1, ops1 = engine.align(params, req, pat, txt)
    assert np.array_equal(res1, res2) and (not bt or engine.format_output(res1, ops1, True) == engine.format_output(res2, ops2, True))


@pytest.mark.parametrize("algo", ["nw", "swg"])
@pytest.mark.parametrize("l,n", [(1000, 96), (400, 300), (2500, 24)])
def test_long_pattern_tails_wrap_the_flat_table_more_than_once(gpu, algo, l, n):
    """Round 6: plen > 2 tlen -- the last row's tail cells v = W .. plen reach back W cells into the tail itself (flat index W tlen + v - W), and
    the walk passes canonical rows beyond tlen + 1. dp_strip_kernel computes them in its tail loop (they join the row's LDS image; their direction bits sit at row
    tlen + v / W, column v mod W) instead of handing the pair to one lane and a pooled table. Texts cut to a third, a fifth, a fiftieth of the pattern and to one
    character, next to ordinary pairs; READ_SIZE 2576 and SWG with CIGAR at READ_SIZE 1040 reach dp_strip_kernel, READ_SIZE 416 and the other READ_SIZE-1040 shapes
    dp_group_kernel, whose tail walk is the same code. Scores and CIGARs against the oracle."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes(algo, l, 0.03)
    req, pat, txt = engine.gen_pairs(6600 + l, 0, n, l, 0.03, rs)
    for i in range(n):
        pl = int(req["pattern_len"][i])
        if i % 4 == 1: req["text_len"][i] = max(1, pl // 3)
        elif i % 4 == 2: req["text_len"][i] = max(1, pl // (5 if i % 8 == 2 else 50))
        elif i % 16 == 3: req["text_len"][i] = 1 + (i // 16) % 3
    for bt in (True, False):
        params = engine.make_params(algo, ms, rs, backtrace=bt, swg_w16=(algo == "swg"))
        _compare(algo, params, req, pat, txt)
    params = engine.make_params(algo, ms, rs, backtrace=True, mismatch=2, **(dict(gap_i=5, gap_d=1) if algo == "nw" else dict(gap_o=1, gap_e=2, swg_w16=True)))
    _compare(algo, params, req, pat, txt)


@pytest.mark.parametrize("rs,l,costs", [(7904, 7800, {}), (4560, 4500, dict(mismatch=7, gap_i=5, gap_d=3)), (4096, 4000, {})])
def test_nw_long_reads_below_the_int16_bound_leave_the_literal_path(gpu, rs, l, costs):
    """Round 6: dp_wave_exact_ok bounded an NW cell by the gap-only path, (2 READ_SIZE + 4) g, and at the launchers' costs sent every pair from READ_SIZE 3 998 on to
    the literal one-lane path (4 GCUPS). A cell is at most READ_SIZE x max(min(x, gi + gd), gi, gd) + g (dp_wave.hpp), so nothing wraps before READ_SIZE 7 990: these
    run on dp_strip_kernel now. Pairs chosen to reach the bound -- unrelated sequences, all-mismatch diagonals, one-character and half-length texts (the flat table's
    aliasing over many rows), a one-character pattern -- next to related ones; READ_SIZE 7 904 is 96 below the first size at which the new bound refuses. Against the
    oracle's int16 table (oracle/aim_oracle.c nw_pair: every store through the reference's casts)."""
    import ctypes as C
    from aim_amd import capi, engine
    n = 10
    req, pat, txt = engine.gen_pairs(9100 + rs, 0, n, l, 0.01, rs)
    rng = np.random.default_rng(rs)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    full = rs - 8
    def put(i, p, t):
        pat[i, :] = 0; txt[i, :] = 0
        pat[i, :len(p)] = p; txt[i, :len(t)] = t
        req["pattern_len"][i] = len(p); req["text_len"][i] = len(t)
    put(0, acgt[rng.integers(0, 4, full)], acgt[rng.integers(0, 4, full)])                    # unrelated, full length
    put(1, np.full(full, ord("A"), np.uint8), np.full(full, ord("C"), np.uint8))              # every diagonal step a mismatch
    put(2, acgt[rng.integers(0, 4, full)], acgt[rng.integers(0, 4, 1)])                       # W = 2: the pattern's row wraps the flat table thousands of times
    put(3, acgt[rng.integers(0, 4, 1)], acgt[rng.integers(0, 4, full)])
    put(4, acgt[rng.integers(0, 4, full)], acgt[rng.integers(0, 4, full // 2)])               # plen = 2 tlen
    put(5, np.full(full, ord("A"), np.uint8), np.concatenate([np.full(full // 3, ord("A"), np.uint8), np.full(full // 4, ord("G"), np.uint8)]))
    put(6, np.full(full // 5, ord("T"), np.uint8), np.full(full, ord("G"), np.uint8))         # plen << tlen, nothing matches
    for bt in (False, True):
        params = engine.make_params("nw", 4, rs, backtrace=bt, **costs)
        assert capi.load().aim_kernel_name(C.byref(params)) == b"dp_strip_kernel"
        _compare("nw", params, req, pat, txt)


def test_dp_strip_batches_with_long_pattern_tails_are_stable_launch_to_launch(gpu):
    """Round 5 filled pairs with plen > 2 tlen on ONE lane out of a small POOL of int16 tables shared by all workgroups under a lock, and 7 % of such batches came back
    wrong until the lock got a device-scope release / acquire (XCD L2s are not coherent inside a kernel; NOTES R5.5 -- this was the regression test). Round 6 computes
    those pairs' tail cells in the strip kernel itself: no pool, no lock, nothing shared between workgroups. The batch that showed the race (READ_SIZE 1032 through
    dp_group's to-do list into dp_strip: every ninth pair's text cut below half its pattern) must give the oracle's output on every one of 10 launches."""
    import random
    from aim_amd import engine
    from oracle import oracle
    for algo, kw, ms in (("swg", dict(swg_w16=True), 0), ("nw", dict(), 40)):
        l, e, rs, n = 948, 0.01, 1032, 445
        params = engine.make_params(algo, ms, rs, backtrace=True, **kw)
        rng = random.Random(5)
        req, pat, txt = engine.gen_pairs(777, 0, n, l, e, rs)
        for i in range(0, n, 9):
            req["text_len"][i] = rng.randint(200, int(req["pattern_len"][i]) // 2 - 1)
        ores, oops, _ = oracle.align_batch(_oracle_params(oracle, params, algo), req["pattern_len"], req["text_len"], pat, txt, nthreads=16)
        want = engine.format_output(ores, oops, True)
        for rep in range(10):
            res, ops = engine.align(params, req, pat, txt, check=False)
            assert np.array_equal(res["status"], ores["status"]) and np.array_equal(res["score"], ores["score"]), (algo, rep)
            assert engine.format_output(res, ops, True) == want, (algo, rep)


@pytest.mark.parametrize("algo,l,err,kw", [("nw", 100, 0.05, {}), ("swg", 100, 0.05, {}), ("swg", 100, 0.02, dict(swg_w16=True)), ("nw", 150, 0.02, {}),
                                           ("swg", 150, 0.02, dict(swg_w16=True)), ("nw", 70, 0.05, {})])
def test_register_kernels_many_groups_per_wavefront(gpu, monkeypatch, algo, l, err, kw):
    """nw_reg_kernel / swg_reg_kernel take groups of 64 candidate pairs through LDS queues (two of them in swg_reg: plen <= tlen / plen > tlen) and run on 64 QUEUED pairs at a
    time; the remainder of a queue moves to its front. With the resident grid of a whole chip the batches of the other tests give a wavefront one group: here the plan is made for
    one CU (AIM_CHIP_CUS=1: 8 wavefronts), so every wavefront drains ~18 groups with tails and short outliers in between."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_CHIP_CUS", "1")
    ms, rs = engine.launcher_sizes(algo, l, err)
    n = 9000
    req, pat, txt = engine.gen_pairs(31 + l, 0, n, l, err, rs)
    rng = np.random.default_rng(l)
    for i in range(0, n, 7):
        req["text_len"][i] = max(1, int(req["text_len"][i]) - int(rng.integers(0, 14)))
    for i in range(3, n, 11):
        req["pattern_len"][i] = max(1, int(req["pattern_len"][i]) - int(rng.integers(0, 30)))
    for bt in (False, True):
        params = engine.make_params(algo, ms, rs, backtrace=bt, **kw)
        _compare(algo, params, req, pat, txt)
        with engine.DeviceSet(1) as s:
            s.configure(params, n)
            assert s.plan_describe(0).startswith(algo + "_reg_kernel") and "grid=8 " in s.plan_describe(0), s.plan_describe(0)


# ------------------------------------------------------------------ medium reads: G lanes per pair (dp_group.hpp, round 5)
@pytest.mark.parametrize("algo", ["nw", "swg"])
@pytest.mark.parametrize("bt", [False, True])
@pytest.mark.parametrize("l,err", [(180, 0.02), (200, 0.05), (250, 0.02), (250, 0.10), (300, 0.05), (320, 0.02), (400, 0.05), (500, 0.02), (700, 0.05), (960, 0.03), (990, 0.02),
                                   (1000, 0.05), (1200, 0.02), (1450, 0.02), (1500, 0.01), (1700, 0.02), (1900, 0.03), (2000, 0.02)])
def test_dp_group_kernel_medium_reads(gpu, monkeypatch, algo, bt, l, err):
    """dp_group_kernel (READ_SIZE 177 .. 1024: G consecutive lanes own a pair, dp_strip's packed row body with the prefix minimum as ONE wave scan over
    (pair rank, value) keys) against the oracle: every length relation -- plen < / == / > tlen incl. long tails (the aliased boundary cell of every row,
    the last row's tail cells incl. plen > 2 tlen (round 6); nw.c:109-153, swg.c:121-171 over the flat table) --, the pairs it leaves to its to-do list (empty
    sequences) through both fallbacks (nw_lane / swg_lane up to READ_SIZE 320, dp_strip in to-do mode above), non-ACGT bytes, other costs (NW: GAP_I != GAP_D),
    and equality with the kernels it replaced (AIM_NO_DP_GROUP=1). Round 6: score-only READ_SIZE 1025 .. 1792 (NW; SWG .. 1536: 20 / 24 / 28 registers per lane, two pairs per
    wavefront; the registers per lane follow lane use x residency at every READ_SIZE), with CIGAR NW to READ_SIZE 1280 (20 registers) and READ_SIZE 1440 .. 2048 (one pair of 45 .. 64 lanes per wavefront); the shapes in between stay on dp_strip_kernel."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes(algo, l, err)
    n = 1500 if l <= 400 else (400 if l <= 1000 else 150)
    req, pat, txt = engine.gen_pairs(8800 + l, 0, n, l, err, rs)
    for i in range(3, n, 17):                                                   # tails of every size
        req["text_len"][i] = max(1, int(req["text_len"][i]) - (i % 61))
    for i in range(5, n, 29):
        req["pattern_len"][i] = max(1, int(req["pattern_len"][i]) - (i % 43))
    req["pattern_len"][11] = min(rs, l); req["text_len"][11] = l // 2 + 3       # plen just below 2 tlen
    req["pattern_len"][12] = 0                                                  # the to-do list: empty sequences, plen > 2 tlen
    req["text_len"][13] = 0
    req["pattern_len"][14] = l; req["text_len"][14] = l // 3
    req["pattern_len"][15] = 1; req["text_len"][15] = 1
    req["pattern_len"][16] = 2; req["text_len"][16] = min(rs, l)
    pat[20, l // 2] = ord("N"); txt[21, l // 3] = ord("N")
    d = req["pattern_len"].astype(int) - req["text_len"].astype(int)
    assert (d < 0).any() and (d == 0).any() and (d == 1).any() and (d >= 2).any() and (d > 32).any()
    costs = ((dict(), dict(mismatch=2, gap_i=5, gap_d=3)) if algo == "nw" else (dict(swg_w16=True), dict(swg_w16=True, mismatch=5, gap_o=2, gap_e=3)))
    for cost in costs:
        params = engine.make_params(algo, ms, rs, backtrace=bt, **cost)
        _compare(algo, params, req, pat, txt)
        with engine.DeviceSet(1) as s:
            s.configure(params, n)
            s.push(0, req, pat, txt); s.launch(); s.pull(0, check=False)
            group = rs <= 1024 or ((algo == "nw" and rs <= 1280) or 1440 <= rs <= 2048 if bt else rs <= (1792 if algo == "nw" else 1536))
            if not group:
                assert s.plan_describe(0).startswith("dp_strip_kernel"), s.plan_describe(0)
                continue
            assert s.plan_describe(0).startswith("dp_group_kernel"), s.plan_describe(0)
            out = int(((req["pattern_len"] < 1) | (req["text_len"] < 1)).sum())
            assert s.fallback_pairs(0) == out and out >= 2, (s.fallback_pairs(0), out)
    params = engine.make_params(algo, ms, rs, backtrace=bt, **costs[0])
    res1, ops1 = engine.align(params, req, pat, txt, check=False)
    monkeypatch.setenv("AIM_NO_DP_GROUP", "1")
    res2, ops2 = engine.align(params, req, pat, txt, check=False)
    assert np.array_equal(res1, res2) and (not bt or engine.format_output(res1, ops1, True) == engine.format_output(res2, ops2, True))


def test_dp_group_kernel_every_lane_count_and_few_pairs(gpu):
    """Every lanes-per-pair value G = 6 .. 32 (READ_SIZE 184 .. 1024 in steps of 32: 10 .. 2 pairs per wavefront, idle lanes behind the last pair), with
    fewer pairs than a wavefront holds and a last unit that is not full, NW and SWG with CIGAR."""
    from aim_amd import engine
    for rs in range(184, 1025, 56):
        l = (rs - 8) * 100 // 104
        for algo, kw in (("nw", dict()), ("swg", dict(swg_w16=True))):
            for n in (1, 64 // ((rs + 31) // 32) + 1, 37):
                req, pat, txt = engine.gen_pairs(100 + rs + n, 0, n, l, 0.03, rs)
                params = engine.make_params(algo, 60, rs, backtrace=True, **kw)
                _compare(algo, params, req, pat, txt)


@pytest.mark.parametrize("algo", ["nw", "swg"])
def test_dp_group_kernel_many_units_per_wavefront(gpu, monkeypatch, algo):
    """A persistent wavefront takes one unit of 64 / G pairs after the other: its LDS slots, its slabs of direction bits and the traceback's window are reused.
    The batches of the other tests fit the resident grid in one unit per wavefront; here the plan is made for a device of 2 CUs (AIM_CHIP_CUS: 16 wavefronts), so every
    wavefront works through a dozen units with pairs of different lengths, tails and outliers in between."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_CHIP_CUS", "2")
    for l, err, n in ((250, 0.03, 1200), (420, 0.02, 700), (900, 0.02, 300)):
        ms, rs = engine.launcher_sizes(algo, l, err)
        req, pat, txt = engine.gen_pairs(660 + l, 0, n, l, err, rs)
        rng = np.random.default_rng(l)
        for i in range(0, n, 3):                                               # lengths all over the row, tails of every size, a few outliers
            req["pattern_len"][i] = int(rng.integers(max(1, l // 3), l))
        for i in range(1, n, 5):
            req["text_len"][i] = int(rng.integers(max(1, l // 2), l))
        req["text_len"][7] = 0; req["pattern_len"][8] = l; req["text_len"][8] = l // 4
        kw = dict(swg_w16=True) if algo == "swg" else dict()
        for bt in (False, True):
            params = engine.make_params(algo, ms, rs, backtrace=bt, **kw)
            _compare(algo, params, req, pat, txt)
            with engine.DeviceSet(1) as s:
                s.configure(params, n)
                assert s.plan_describe(0).startswith("dp_group_kernel") and "grid=16 " in s.plan_describe(0), s.plan_describe(0)


# ------------------------------------------------------------------ NW with GAP_I != GAP_D
@pytest.mark.parametrize("gi,gd,mism", [(2, 7, 3), (7, 3, 5), (1, 6, 2)])
@pytest.mark.parametrize("bt", [False, True])
@pytest.mark.parametrize("kernel", ["nw_reg", "nw_lane", "dp_strip", "dp_wave"])
def test_nw_asymmetric_gap_costs_on_every_nw_kernel(gpu, monkeypatch, kernel, bt, gi, gd, mism):
    """nw.c:67-153 takes GAP_I (move along the pattern) and GAP_D (move along the text) as two macros; the launchers set them equal, the
    ABI does not have to (aim_hip.h aim_params_t.gap_i / gap_d). nw_reg_kernel's tilted coordinates T = R - GAP_I*h - GAP_D*v are exactly
    where the two stop being interchangeable: all four NW kernels against the oracle with the costs apart, all three length relations
    (plen < / == / > tlen, incl. the aliased tails plen >= tlen + 2), score-only and with CIGAR."""
    from aim_amd import engine
    l, err, n = (100, 0.05, 4000) if kernel in ("nw_reg", "nw_lane") else (1000, 0.05, 160)
    ms, rs = engine.launcher_sizes("nw", l, err)
    if kernel == "nw_lane":
        monkeypatch.setenv("AIM_NO_NW_REG", "1")
    if kernel == "dp_wave":
        monkeypatch.setenv("AIM_DPW_LEGACY", "1")
    if kernel == "dp_strip":
        monkeypatch.setenv("AIM_NO_DP_GROUP", "1")                             # (round 6: score-only READ_SIZE 1064 is dp_group_kernel's, 20 registers per lane)
    req, pat, txt = engine.gen_pairs(9100 + gi, 0, n, l, err, rs)
    for i in range(3, n, 41):                                                  # more tails / short texts than the generator draws
        req["text_len"][i] = max(1, int(req["text_len"][i]) - (i % 9))
    for i in range(7, n, 53):
        req["pattern_len"][i] = max(1, int(req["pattern_len"][i]) - (i % 11))
    d = req["pattern_len"].astype(int) - req["text_len"].astype(int)
    assert (d < 0).any() and (d == 0).any() and (d == 1).any() and (d >= 2).any()
    params = engine.make_params("nw", ms, rs, backtrace=bt, mismatch=mism, gap_i=gi, gap_d=gd)
    _compare("nw", params, req, pat, txt)
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        s.push(0, req, pat, txt); s.launch(); s.pull(0)
        assert s.plan_describe(0).startswith(kernel + "_kernel"), s.plan_describe(0)


def test_host_cli_takes_gap_i_and_gap_d(gpu, tmp_path):
    """judge r04 row 1 through the host CLI (--gap-i / --gap-d): the reference's output digest for GAP_I 2, GAP_D 7."""
    case = [c for c in judge_cases() if c["name"] == "nw_asym_gi2_gd7_l100_e5_bt"][0]
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(judge_case_input(case))
    r = _host_cli(case, inp, out, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert md5(out.read_bytes()) == case["output_md5"]


# ------------------------------------------------------------------ swg_reg_kernel
@pytest.mark.parametrize("l,err", [(100, 0.01), (100, 0.05), (100, 0.10), (70, 0.02), (60, 0.10), (90, 0.03), (104, 0.0), (40, 0.05), (54, 0.02), (150, 0.01), (150, 0.05),
                                   (130, 0.03), (165, 0.02)])
@pytest.mark.parametrize("bt", [False, True])
def test_swg_rows_in_registers_kernel(gpu, monkeypatch, l, err, bt):
    """swg_reg_kernel (dp_reg.hpp: M and I rows in registers, left-aligned, value * 256 in 16-bit fields for int8 cells) against the oracle: both
    length relations it takes (plen <= tlen, plen == tlen + 1: the aliased boundary cell {M, D}), the pairs it hands to swg_lane_kernel (plen >= tlen + 2,
    short outliers, pairs whose int8 cells wrap), every READ_SIZE class (48 ... 128), int8 and int16 cells, other penalties, and equality with the
    LDS-row kernel alone (AIM_NO_SWG_REG=1)."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes("swg", l, max(err, 0.01))
    n = 6000
    req, pat, txt = engine.gen_pairs(8000 + l, 0, n, l, err, rs)
    for i in range(0, n, 97):                                                  # outliers: a few short pairs, a few long tails, unrelated texts
        req["pattern_len"][i] = max(1, l // 3)
    for i in range(5, n, 131):
        req["text_len"][i] = max(1, int(req["text_len"][i]) - 7)
    rng = np.random.default_rng(l)
    for i in range(11, n, 61):                                                 # unrelated sequences: cells reach MAX_SCORE + min(h, v) e (quirk S2) and int8 cells wrap
        txt[i, :int(req["text_len"][i])] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(req["text_len"][i]))
    for cost, kw in ((dict(), dict()), (dict(mismatch=2, gap_o=5, gap_e=1), dict()), (dict(mismatch=7, gap_o=3, gap_e=2), dict()),
                     (dict(), dict(swg_w16=True)), (dict(mismatch=5, gap_o=2, gap_e=3), dict(swg_w16=True))):
        params = engine.make_params("swg", ms, rs, backtrace=bt, **cost, **kw)
        res, ops, _ = _compare("swg", params, req, pat, txt, expect_status=None)
        with engine.DeviceSet(1) as s:
            s.configure(params, n)
            s.push(0, req, pat, txt); s.launch(); s.pull(0, check=False)
            int8_wraps = not kw and cost.get("gap_o", 4) + (rs - 8) * cost.get("gap_e", 1) > 127       # the column initialisation o + v e passes 127: every pair wraps (S3)
            if int8_wraps:
                assert s.plan_describe(0).startswith("swg_lane_kernel"), s.plan_describe(0)
                continue
            assert s.plan_describe(0).startswith("swg_reg_kernel"), s.plan_describe(0)
            fb = s.fallback_pairs(0)
            tails = int((req["pattern_len"] > req["text_len"] + 9).sum())     # more than 8 tail cells in the last row
            assert tails <= fb, (fb, tails)
            if not cost and not kw and err <= 0.05 and l >= 60:
                assert fb <= tails + n // 20, (fb, tails)                       # default costs: the to-do list is the tail pairs + the outliers + the unrelated pairs
    monkeypatch.setenv("AIM_NO_SWG_REG", "1")
    params = engine.make_params("swg", ms, rs, backtrace=bt, swg_w16=l >= 120)
    res2, ops2 = engine.align(params, req, pat, txt, check=False)
    monkeypatch.delenv("AIM_NO_SWG_REG")
    res1, ops1 = engine.align(params, req, pat, txt, check=False)
    assert np.array_equal(res1, res2) and (not bt or engine.format_output(res1, ops1, True) == engine.format_output(res2, ops2, True))


@pytest.mark.parametrize("algo,l", [("nw", 100), ("swg", 100), ("nw", 150), ("swg", 150), ("nw", 60), ("swg", 40)])
def test_register_kernels_walk_that_leaves_the_band_goes_to_the_todo_list(gpu, algo, l):
    """Round 6: with CIGAR nw_reg_kernel / swg_reg_kernel keep direction bits only for a band of four dwords (64 / 32 columns) around the wavefront's centre
    line, fetch them eight rows at a time, and send a pair whose walk leaves the band to the to-do list (nw_lane / swg_lane), as dp_strip does for cfg4. With
    cheap gaps a text that is the pattern shifted by a third of its length (a block missing in front, other bases appended: lengths stay equal) is aligned
    along a diagonal ~l/3 off the corner-to-corner one; related pairs in the same wavefronts stay inside. Scores and CIGARs against the oracle, both kinds; the
    to-do list holds shifted pairs only."""
    from aim_amd import engine
    ms, rs = engine.launcher_sizes(algo, l, 0.02)
    n = 4000
    req, pat, txt = engine.gen_pairs(6100 + l, 0, n, l, 0.02, rs)
    shift = max(l * 45 // 100, 21)                                            # the window reaches at most 40 (NW) / 20 (SWG) columns to one side of the centre line
    shifted = np.zeros(n, dtype=bool)
    for i in range(0, n, 3):
        # pattern = A^shift + B, text = B + C^shift (B random): the only long common subsequence is B, `shift` diagonals off the corner-to-corner line -- random flanks
        # would let a gap-cheap alignment wander along the main diagonal instead (the LCS of two random reads is ~0.65 l)
        pl = int(req["pattern_len"][i])
        pat[i, :shift] = ord("A")
        t = np.concatenate([pat[i, shift:pl], np.full(shift, ord("C"), dtype=np.uint8)])
        txt[i, :] = 0
        txt[i, :pl] = t
        req["text_len"][i] = pl
        shifted[i] = True
    cost = dict(mismatch=5, gap=1) if algo == "nw" else dict(mismatch=6, gap_o=1, gap_e=1)
    kw = dict(swg_w16=True) if algo == "swg" else {}
    params = engine.make_params(algo, 60 if algo == "swg" else ms, rs, backtrace=True, **cost, **kw)
    res, ops, ores = _compare(algo, params, req, pat, txt)
    assert np.median(ores["score"][shifted]) <= 2 * shift + 2 + 2 * l // 20     # the shifted pairs ARE aligned through the gaps (2 * shift gap columns + their own few edits), not along the main diagonal
    with engine.DeviceSet(1) as s:
        s.configure(params, n)
        s.push(0, req, pat, txt); s.launch(); s.pull(0)
        assert s.plan_describe(0).startswith(algo + "_reg_kernel"), s.plan_describe(0)
        fb = s.fallback_pairs(0)
    assert fb <= shifted.sum() + n // 50, (fb, int(shifted.sum()))
    if shift > (40 if algo == "nw" else 20):
        assert fb >= 0.8 * shifted.sum(), (fb, int(shifted.sum()))
    # the same batch score-only never needs the band
    params = engine.make_params(algo, 60 if algo == "swg" else ms, rs, backtrace=False, **cost, **kw)
    _compare(algo, params, req, pat, txt)


def test_swg_register_kernel_sends_wrapping_pairs_to_the_literal_kernel(gpu):
    """int8 cells (MAX_SCORE 25, l = 100): unrelated sequences drive cells past 127 (S2 + S3); the register kernel must notice (sign of the OR of
    every stored M) and leave exactly such pairs to swg_lane_kernel -- bit-exact either way, incl. AIM_PAIR_SWG_NO_OP where the oracle has it."""
    from aim_amd import capi, engine
    rng = np.random.default_rng(5)
    n = 2000
    req, pat, txt = engine.gen_pairs(31, 0, n, 100, 0.05, 112)
    for i in range(0, n, 2):
        txt[i, :int(req["text_len"][i])] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(req["text_len"][i]))
    for bt in (False, True):
        params = engine.make_params("swg", 25, 112, backtrace=bt)
        _compare("swg", params, req, pat, txt)
        with engine.DeviceSet(1) as s:
            s.configure(params, n)
            s.push(0, req, pat, txt); s.launch(); s.pull(0, check=False)
            assert s.plan_describe(0).startswith("swg_reg_kernel")
            assert n // 4 <= s.fallback_pairs(0) <= n // 2 + n // 5           # the unrelated half (those that wrap) and the related pairs whose cells pass 127 far from the diagonal -- not everything


# ------------------------------------------------------------------ lane kernels beyond 3 / 4 / 1
LANE_COSTS = [dict(mismatch=4, gap_o=6, gap_e=2), dict(mismatch=2, gap_o=3, gap_e=1), dict(mismatch=5, gap_o=4, gap_e=2)]


@pytest.mark.parametrize("cost", LANE_COSTS, ids=lambda c: "x%dg%da%d" % (c["mismatch"], c["gap_o"], c["gap_e"]))
@pytest.mark.parametrize("l,rs", [(100, 112), (70, 80), (150, 160)])
@pytest.mark.parametrize("bt", [False, True])
def test_lane_kernels_take_the_launchers_other_penalty_sets(gpu, cost, l, rs, bt):
    """run-wfa-pim-wram.py:17-24 takes any -x -g -a; the one-pair-per-lane kernels are built for the sets of AIM_LANE_COST_SETS (wfa_lane.hpp) at the
    launcher's MAX_SCORE for e = 1 % (ceil(l e) max(x, o + e)): ASCII rows through the default ABI (wfa_lane_kernel at READ_SIZE 80 / 112,
    wfa_lane_packed_kernel after pack_rows elsewhere) and packed batches with the compact CIGAR -- against the oracle, incl. pairs beyond the
    cap, a run-time MAX_SCORE below the shape's, and pairs with bytes outside A/C/G/T."""
    from aim_amd import capi, engine
    import ctypes as C
    cap = {4: 8, 2: 4, 5: 6}[cost["mismatch"]]                           # MAX_SCORE of the set's static shape (AIM_LANE_COST_SETS)
    ms, rs_l = engine.launcher_sizes("wfa", l, 0.01, **cost)
    assert rs_l == rs
    ms = min(ms, cap)                                                    # (l = 150: the launcher's MAX_SCORE is two edits' worth -- the group kernel's)
    n = 4000 + 21
    req, pat, txt = engine.gen_pairs(900 + l, 0, n, l, 0.01, rs)
    r2, p2, t2 = engine.gen_pairs(901 + l, 0, 400, l, 0.03, rs)          # some pairs beyond MAX_SCORE
    req[1000:1400], pat[1000:1400], txt[1000:1400] = r2, p2, t2
    req["idx"] = np.arange(n)
    for i in range(0, n, 97):
        pat[i, i % (l // 2)] = ord("N")
    for ms_run in (ms, max(0, ms - cost["gap_e"] - 1)):
        for reduce in (True, False):
            params = engine.make_params("wfa", ms_run, rs, backtrace=bt, reduce=reduce, **cost)
            want = b"wfa_lane_kernel" if rs in (80, 112) else b"wfa_lane_packed_kernel"
            assert capi.load().aim_kernel_name(C.byref(params)) == want
            res, _, ores = _compare("wfa", params, req, pat, txt)
            assert (ores["score"] <= ms_run).any() and (ores["score"] == ms_run + 1).any()
    params = engine.make_params("wfa", ms, rs, backtrace=bt, reduce=True, req8=True, res8=not bt, **cost)
    ores, want_text = _oracle_text("wfa", params, req, pat, txt)
    out, _ = _fused(params, req, pat, txt)
    if bt:
        assert np.array_equal(out["cig"]["score"], ores["score"]) and engine.format_output_runs(out["cig"], out["runs"]) == want_text
    else:
        assert np.array_equal(out["res"]["score"], ores["score"])
    bigger = engine.make_params("wfa", cap + 1, rs, backtrace=bt, **cost)    # beyond the static shape: the group kernel
    assert capi.load().aim_kernel_name(C.byref(bigger)) == b"wfa_group_kernel"


def test_default_abi_ops_rows_need_no_prefilled_buffer(gpu, sample_bytes, ref_digests, monkeypatch):
    """the kernels write 'M' only where an operation can be printed (ops[begin_offset, end_offset), host.c:347-349) -- whatever the
    caller's ops buffer held before (here: 0xEE everywhere, through aim_align_device-style reuse of one device set) the reference's file comes out."""
    from aim_amd import engine
    monkeypatch.setenv("AIM_DEBUG_POISON_OPS", "238")                      # every launch starts from rows of 0xEE
    req, pat, txt = engine.parse_pairs(sample_bytes, 112)
    n = len(req)
    with engine.DeviceSet(1) as s:
        for algo, ms, key, kw in (("swg", 5, "swg_w8_backtrace", {}), ("wfa", 5, "wfa_reduce_backtrace", dict(reduce=True)), ("nw", 4, "nw_backtrace", {}),
                                  ("wfa", 5, "wfa_backtrace", {})):
            # poison: a first launch whose CIGARs are long (unrelated texts) fills the device rows with other operations
            rng = np.random.default_rng(3)
            junk = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=txt.shape).astype(np.uint8)
            pj = engine.make_params(algo, 200 if algo != "wfa" else 5, 112, backtrace=True, **kw)
            s.configure(pj, n)
            s.push(0, req, pat, junk); s.launch(); s.pull(0, check=False)
            params = engine.make_params(algo, ms, 112, backtrace=True, **kw)
            s.configure(params, n)
            s.push(0, req, pat, txt); s.launch()
            res, ops = s.pull(0)
            assert md5(engine.format_output(res, ops, True)) == ref_digests[key]


@pytest.mark.parametrize("algo", ["swg", "nw"])
def test_dp_strip_walk_that_leaves_the_band_is_redone_with_every_strips_bits(gpu, monkeypatch, algo):
    """Round 5: with CIGAR the K = 20 strip shape makes direction bits only in the strips around the diagonal and fills a pair again when its walk leaves
    that band (dp_strip.hpp). Pairs whose optimal path shifts by ~900 diagonals half-way (a 900-base block missing from the text, 900 other bases appended: the
    lengths stay equal) leave it; related pairs in the same batch do not. Both against the oracle, scores and CIGARs."""
    from aim_amd import capi, engine
    import ctypes as C
    monkeypatch.setenv("AIM_STRIP_K", "20")
    l, rs, n = 3000, 3064, 24
    ms = 500 if algo == "swg" else 400
    req, pat, txt = engine.gen_pairs(515, 0, n, l, 0.01, rs)
    rng = np.random.default_rng(8)
    for i in range(0, n, 2):
        p = pat[i, :l].copy()
        t = np.concatenate([p[:1000], p[1900:], rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=900).astype(np.uint8)])
        txt[i, :] = 0
        txt[i, :l] = t
        req["pattern_len"][i] = l
        req["text_len"][i] = l
    params = engine.make_params(algo, ms, rs, backtrace=True, swg_w16=(algo == "swg"))
    assert capi.load().aim_kernel_name(C.byref(params)) == b"dp_strip_kernel"
    res, ops, ores = _compare(algo, params, req, pat, txt)
    assert (ores["score"][0::2] > 4 * ores["score"][1::2].max()).all()          # the shifted pairs really are different animals
