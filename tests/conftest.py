import gzip
import hashlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces built in-tree (no-op when up to date)."""
    from aim_amd import build
    build.build_all()
    from oracle import oracle
    oracle.build()
    return True


@pytest.fixture(scope="session")
def sample_bytes():
    with gzip.open(os.path.join(GOLDEN, "sample-l100-e1-40K.gz"), "rb") as f:
        data = f.read()
    digests = json.load(open(os.path.join(GOLDEN, "reference_digests.json")))
    assert hashlib.md5(data).hexdigest() == digests["input_md5"]
    return data


@pytest.fixture(scope="session")
def err_full_bytes():
    """The reference's own real-read set Datasets/ERR240727-l100-e1-30000Pairs (15 000 pairs, contains 'N'), whole file."""
    with gzip.open(os.path.join(GOLDEN, "ERR240727-l100-e1-30000Pairs.gz"), "rb") as f:
        data = f.read()
    digests = json.load(open(os.path.join(GOLDEN, "reference_digests.json")))
    assert hashlib.md5(data).hexdigest() == digests["err240727_input_md5"]
    return data


@pytest.fixture(scope="session")
def err_bytes(err_full_bytes):
    """Its first 2 000 pairs."""
    return b"\n".join(err_full_bytes.split(b"\n")[:4000]) + b"\n"


@pytest.fixture(scope="session")
def ref_digests():
    return json.load(open(os.path.join(GOLDEN, "reference_digests.json")))


def md5(b):
    return hashlib.md5(b).hexdigest()


def judge_cases():
    """The wider reference digests the judges recorded: 14 in round 1 (tests/golden/judge_r01_cases.json), 14 in round 2
    (judge_r02_cases.json: dynamic-bounds / READ_SIZE-80 lane shapes, l = 150, MAX_SCORE 10, MRAM-variant overflow) and 20 in
    round 3 (judge_r03_cases.json: score-unit loop 4/6/2 and 6/2/2, READ_SIZE 136 ... 184 with CIGAR, rows of 96, NW / SWG at
    cfg4's real size) and 13 in round 4 (judge_r04_cases.json: NW with GAP_I != GAP_D, the READ_SIZE 80 / 128 register shapes,
    WFA-adaptive at l = 10 000 / 4 000 / 2 000) and 31 in round 5 (judge_r05_cases.json: dp_group_kernel's lengths, swg_reg with
    other costs, READ_SIZE 144 ... 176, NW bits on dp_strip, the lane kernels' other penalty sets, and eight `tails_v1` inputs --
    long tails and the literal path). Rows where the reference aborts are in judge_abort_cases()."""
    cases = []
    for name in ("judge_r01_cases.json", "judge_r02_cases.json", "judge_r03_cases.json", "judge_r04_cases.json",
                 "judge_r05_cases.json"):
        cases += [c for c in json.load(open(os.path.join(GOLDEN, name)))["cases"] if "abort" not in c]
    return cases


def judge_abort_cases():
    """judge r03: inputs on which the reference AND the oracle stop with `SWG backtrace. No backtrace operation found`, exit 1
    (swg.c:99-104: int8 cells wrapped on store)."""
    return [c for c in json.load(open(os.path.join(GOLDEN, "judge_r03_cases.json")))["cases"] if "abort" in c]


def judge_dataset_cases():
    """judge r03: reference digests on the reference's own Datasets/ERR240727-l100-e1-30000Pairs, NR_DPUS 4, <n> 15000."""
    return json.load(open(os.path.join(GOLDEN, "judge_r03_cases.json")))["dataset_cases"]


def judge_costs(case):
    return {k: case[k] for k in ("mismatch", "gap_o", "gap_e", "gap", "gap_i", "gap_d") if k in case}


def tails_v1(data):
    """judge r05's transform of a gen_dataset file (Next round item 1): pair i (0-based), pattern p / text t without
    the marker byte, in this order -- i % 5 == 3: t loses its last i % 61 characters; i % 11 == 7: p loses its last i % 43;
    i % 97 == 13: t = t[:len(p) // 3] (never below one character)."""
    lines = data.split(b"\n")[:-1]
    out = []
    for i in range(len(lines) // 2):
        p, t = lines[2 * i][1:], lines[2 * i + 1][1:]
        if i % 5 == 3:
            t = t[:max(1, len(t) - i % 61)]
        if i % 11 == 7:
            p = p[:max(1, len(p) - i % 43)]
        if i % 97 == 13:
            t = t[:max(1, len(p) // 3)]
        out += [lines[2 * i][:1] + p, lines[2 * i + 1][:1] + t]
    return b"\n".join(out) + b"\n"


def judge_case_input(case):
    """Regenerate a judge case's input file exactly as `python -m aim_amd.gen_dataset -n N -l L -e E -s SEED` writes it (and, for
    the rows that name one, the judge's transform of that file)."""
    from aim_amd import engine
    g = case["gen"]
    edits = int(-(-g["l"] * g["e"] // 1))
    row = (g["l"] + edits + 1 + 7) // 8 * 8
    req, pat, txt = engine.gen_pairs(g["s"], 0, g["n"], g["l"], g["e"], row)
    data = engine.pairs_to_text(req, pat, txt)
    if case.get("transform") == "tails_v1":
        data = tails_v1(data)
    else:
        assert "transform" not in case, case["transform"]
    assert hashlib.md5(data).hexdigest() == case["input_md5"], "generator drifted from the judge's input for " + case["name"]
    return data


def reduce_cases():
    """Constructed pairs on which WFA-adaptive's reduction changes the score (tests/golden/reduce_changes_score.json)."""
    import numpy as np
    from aim_amd import engine
    d = json.load(open(os.path.join(GOLDEN, "reduce_changes_score.json")))
    data = b"".join(b">" + c["pattern"].encode() + b"\n<" + c["text"].encode() + b"\n" for c in d["pairs"])
    req, pat, txt = engine.parse_pairs(data, d["read_size"])
    plain = np.array([c["score_plain"] for c in d["pairs"]])
    red = np.array([c["score_reduce"] for c in d["pairs"]])
    return d, req, pat, txt, plain, red
