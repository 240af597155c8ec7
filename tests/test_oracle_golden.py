"""The CPU oracle against everything the reference left us for this path: the recorded output
digests and score histograms on Datasets/sample-l100-e1-40K (SURVEY.md 8a / BASELINE.md 2)."""
import collections
import os
import subprocess

import numpy as np
import pytest

from conftest import (ROOT, judge_abort_cases, judge_case_input, judge_cases, judge_costs, judge_dataset_cases, md5,
                      reduce_cases)


def _run(oracle_mod, engine, data, algo, max_score, backtrace, reduce=False, swg_cell_bytes=0, threads=4):
    req, pat, txt = engine.parse_pairs(data, 112)
    p = oracle_mod.params(algo, max_score, 112, backtrace=backtrace, reduce=reduce, swg_cell_bytes=swg_cell_bytes)
    res, ops, worst = oracle_mod.align_batch(p, req["pattern_len"], req["text_len"], pat, txt, nthreads=threads)
    assert worst == 0
    return res, ops


CASES = [
    ("wfa_backtrace", "wfa", 5, True, False, 0),
    ("wfa_reduce_backtrace", "wfa", 5, True, True, 0),
    ("wfa_score_only", "wfa", 5, False, True, 0),
    ("nw_backtrace", "nw", 4, True, False, 0),
    ("swg_w8_backtrace", "swg", 5, True, False, 1),
    ("swg_w16_backtrace", "swg", 5, True, False, 2),
]


@pytest.mark.parametrize("key,algo,ms,bt,red,cellb", CASES)
def test_oracle_reproduces_reference_digest(built, sample_bytes, ref_digests, key, algo, ms, bt, red, cellb):
    from aim_amd import engine
    from oracle import oracle
    res, ops = _run(oracle, engine, sample_bytes, algo, ms, bt, red, cellb)
    assert len(res) == 20000
    out = oracle.format_output(res, ops, bt)
    assert md5(out) == ref_digests[key]


def test_oracle_score_histograms(built, sample_bytes, ref_digests):
    from aim_amd import engine
    from oracle import oracle
    for algo, ms, key in (("wfa", 5, "score_histogram_wfa_swg"), ("swg", 5, "score_histogram_wfa_swg"),
                          ("nw", 4, "score_histogram_nw")):
        res, _ = _run(oracle, engine, sample_bytes, algo, ms, False)
        hist = collections.Counter(int(s) for s in res["score"])
        assert {str(k): v for k, v in hist.items()} == ref_digests[key]


def test_oracle_cli_file_digest(built, sample_bytes, ref_digests, tmp_path):
    """Whole-program check incl. the restated parser / partition rule / writer (host.c:91-134,191,331-352)."""
    inp = tmp_path / "sample"
    inp.write_bytes(sample_bytes)
    cli = os.path.join(ROOT, "oracle", "oracle_cli")
    for algo, flags, key in (("wfa", ["-b", "-r"], "wfa_reduce_backtrace"), ("nw", ["-b"], "nw_backtrace"),
                             ("wfa", ["-r"], "wfa_score_only")):
        out = tmp_path / ("out_" + key)
        subprocess.check_call([cli, algo, "-i", str(inp), "-o", str(out), "-n", "20000", "-l", "100", "-e", "0.01",
                               "-d", "4", "-t", "4"] + flags)
        assert md5(out.read_bytes()) == ref_digests[key]
    # H3: n is not a cap -- n=100, NR_DPUS=4 -> ROUND_UP_8(25)*4 = 128 pairs
    out = tmp_path / "out_n100"
    subprocess.check_call([cli, "wfa", "-i", str(inp), "-o", str(out), "-n", "100", "-l", "100", "-e", "0.01", "-d", "4"])
    assert out.read_bytes().count(b"\n") == 128
    # H4: n <= NR_DPUS exits 1
    assert subprocess.call([cli, "wfa", "-i", str(inp), "-o", str(out), "-n", "4", "-l", "100", "-e", "0.01", "-d", "4"],
                           stdout=subprocess.DEVNULL) == 1


def test_wfa_wram_equals_mram_style_variants(built, sample_bytes):
    """REDUCE is inert at l=100 (width < 10) and score-only scores equal the backtrace run's scores."""
    from aim_amd import engine
    from oracle import oracle
    a, _ = _run(oracle, engine, sample_bytes, "wfa", 5, True, False)
    b, _ = _run(oracle, engine, sample_bytes, "wfa", 5, False, True)
    assert np.array_equal(a["score"], b["score"])


@pytest.mark.parametrize("case", judge_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_judge_r01_reference_digests(built, case):
    """Wider reference output digests (aliasing, int8 wrap, reduce firing, long reads, MAX_SCORE overflow, custom
    penalties) recorded by the round-1 judge from the reference under a UPMEM shim (judge_r01_cases.json)."""
    from aim_amd import engine
    from oracle import oracle
    data = judge_case_input(case)
    req, pat, txt = engine.parse_pairs(data, case["read_size"])
    p = oracle.params(case["algo"], case["max_score"], case["read_size"], backtrace=case["backtrace"],
                      reduce=case.get("reduce", False), swg_cell_bytes=case.get("swg_cell_bytes", 0), **judge_costs(case))
    res, ops, worst = oracle.align_batch(p, req["pattern_len"], req["text_len"], pat, txt, nthreads=4)
    assert worst == 0 and len(res) == case["gen"]["n"]
    assert md5(oracle.format_output(res, ops, case["backtrace"])) == case["output_md5"]


@pytest.mark.parametrize("case", judge_abort_cases(), ids=lambda c: c["name"])
def test_oracle_aborts_where_the_reference_aborts(built, case, tmp_path):
    """judge r03: the reference stops with `SWG backtrace. No backtrace operation found` + exit(1) (swg.c:99-104) on these inputs
    (int8 cells wrapped on store); so do the oracle and its CLI."""
    from aim_amd import engine
    from oracle import oracle
    data = judge_case_input(case)
    req, pat, txt = engine.parse_pairs(data, case["read_size"])
    p = oracle.params(case["algo"], case["max_score"], case["read_size"], backtrace=True,
                      swg_cell_bytes=case.get("swg_cell_bytes", 0), **judge_costs(case))
    _, _, worst = oracle.align_batch(p, req["pattern_len"], req["text_len"], pat, txt, nthreads=4)
    assert worst != 0
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(data)
    g, c = case["gen"], judge_costs(case)
    r = subprocess.run([os.path.join(ROOT, "oracle", "oracle_cli"), "swg", "-i", str(inp), "-o", str(out), "-n", str(g["n"]),
                        "-l", str(g["l"]), "-e", str(g["e"]), "-b", "--max-score", str(case["max_score"]), "--read-size",
                        str(case["read_size"]), "-x", str(c.get("mismatch", 3)), "-g", str(c.get("gap_o", 4)), "-a",
                        str(c.get("gap_e", 1))], capture_output=True, text=True)
    assert r.returncode == 1 and case["abort"] in r.stdout


@pytest.mark.parametrize("case", judge_dataset_cases(), ids=lambda c: c["name"])
def test_oracle_cli_reproduces_judge_r03_dataset_digests(built, case, err_full_bytes, tmp_path):
    """judge r03: whole-file digests of the reference on its own real-read set, 4 DPUs (oracle_cli restates parser + partition)."""
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(err_full_bytes)
    flags = (["-b"] if case["backtrace"] else []) + (["-r"] if case.get("reduce") else [])
    subprocess.check_call([os.path.join(ROOT, "oracle", "oracle_cli"), case["algo"], "-i", str(inp), "-o", str(out), "-n",
                           str(case["n"]), "-l", "100", "-e", "0.01", "-d", str(case["nr_dpus"]), "-t", "4", "--max-score",
                           str(case["max_score"]), "--read-size", str(case["read_size"])] + flags)
    assert md5(out.read_bytes()) == case["output_md5"]


def test_oracle_reduction_changes_scores_on_constructed_pairs(built):
    """affine_wfa_reduce_wvs (wfa.c:69-140) is a heuristic: it may cut the diagonal the optimum needs. Random reads never show
    it; these constructed pairs do (21 without -r, 22 / 25 with). Pins the oracle's reduction as something that ACTS."""
    from oracle import oracle
    d, req, pat, txt, plain, red = reduce_cases()
    for bt in (False, True):
        a, _, _ = oracle.align_batch(oracle.params("wfa", d["max_score"], d["read_size"], backtrace=bt, reduce=False),
                                     req["pattern_len"], req["text_len"], pat, txt)
        b, _, _ = oracle.align_batch(oracle.params("wfa", d["max_score"], d["read_size"], backtrace=bt, reduce=True),
                                     req["pattern_len"], req["text_len"], pat, txt)
        assert np.array_equal(a["score"], plain) and np.array_equal(b["score"], red)
        assert (plain != red).all()
