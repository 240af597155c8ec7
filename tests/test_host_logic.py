"""Host-side logic that needs no GPU: parser (get_reads), writer / edit_cigar_print, generator, host CLI argument
handling -- compared with the oracle's restatement where one exists."""
import os
import subprocess

import numpy as np

from conftest import ROOT, md5


def test_parse_pairs_follows_get_reads(built):
    from aim_amd import engine
    data = b">ACGT\n<ACGGT\n>AAAA\n<AAA\n>CC\n<CC"          # last line has no newline: loses its final base (H1)
    req, pat, txt = engine.parse_pairs(data, 8)
    assert req["pattern_len"].tolist() == [4, 4, 2] and req["text_len"].tolist() == [5, 3, 1]
    assert pat[0, :4].tobytes() == b"ACGT" and txt[0, :5].tobytes() == b"ACGGT" and txt[2, :1].tobytes() == b"C"
    assert req["idx"].tolist() == [0, 1, 2]
    try:
        engine.parse_pairs(b">ACGTACGTA\n<A\n", 8)
        assert False
    except ValueError as e:
        assert "READ LENGTH" in str(e)


def test_cigar_format_matches_oracle(built):
    from aim_amd import engine
    from oracle import oracle
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(1, 230))
        ops = rng.choice(np.frombuffer(b"MMMMMMXID", dtype=np.uint8), size=n + 5).astype(np.uint8)
        b = int(rng.integers(0, n))
        assert engine.cigar_of(ops, b, n) == oracle.cigar_of(ops, b, n)
    ops = np.frombuffer(b"MMMXMMIIDM", dtype=np.uint8).copy()
    assert engine.cigar_of(ops, 0, 10) == b"3M1X2M2I1D1M\n"
    assert engine.cigar_of(ops, 9, 10) == b"1M\n"


def test_generator_is_deterministic_and_shardable(built):
    from aim_amd import engine
    req, pat, txt = engine.gen_pairs(42, 0, 4096, 100, 0.01, 112)
    digest = md5(req.tobytes() + pat.tobytes() + txt.tobytes())
    assert digest == "1a0b9e39dfaa82173383703e6336a3d8", digest
    # any shard regenerates independently of how the batch is cut
    r2, p2, t2 = engine.gen_pairs(42, 1000, 96, 100, 0.01, 112)
    assert np.array_equal(r2, req[1000:1096]) and np.array_equal(p2, pat[1000:1096]) and np.array_equal(t2, txt[1000:1096])
    # statistics of the recipe: one edit per pair -> lengths 99/100/101 in thirds, substitutions may redraw the base
    d = (req["text_len"] - req["pattern_len"])
    assert set(d.tolist()) == {-1, 0, 1}
    for v in (-1, 0, 1):
        assert 0.28 < (d == v).mean() < 0.39
    same = ((pat == txt).all(axis=1) & (d == 0)).mean()
    assert 0.05 < same < 0.12
    assert set(np.unique(pat[:, :100]).tolist()) == set(b"ACGT")
    r3, _, t3 = engine.gen_pairs(7, 0, 500, 1000, 0.05, 1064)
    assert (np.abs(r3["text_len"] - 1000) <= 50).all() and (t3[np.arange(500), r3["text_len"] - 1] != 0).all()


def test_host_cli_argument_errors(built, tmp_path):
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    r = subprocess.run([host], capture_output=True, text=True)
    assert r.returncode == 1 and "wrong number of arguments" in r.stdout          # host.c:150-154
    inp = tmp_path / "in"
    inp.write_bytes(b">ACGT\n<ACGT\n" * 20)
    r = subprocess.run([host, str(inp), str(tmp_path / "o"), "0"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Invalid nb of reads" in r.stderr                # host.c:175-179
    r = subprocess.run([host, str(inp), str(tmp_path / "o"), "4", "--nr-dpus", "4"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Allocated DPUs more than needed" in r.stdout    # host.c:180-184
    r = subprocess.run([host, str(tmp_path / "missing"), str(tmp_path / "o"), "10"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "couldn't be opened" in r.stderr


def test_launch_dry_run_command(built, capsys):
    from aim_amd import launch
    assert launch.main(["wfa", "-i", "a", "-o", "b", "-l", "100", "-e", "0.01", "-n", "40", "-b", "-r", "-d", "4", "--dry-run"]) == 0
    out = capsys.readouterr().out
    assert "-DMAX_SCORE=5 -DREAD_SIZE=112 -DMATCH=0 -DMISMATCH=3 -DGAP_O=4 -DGAP_E=1 -DREDUCE -DBACKTRACE" in out
    assert "--algo wfa --max-score 5 --read-size 112" in out and "--nr-dpus 4" in out and "--backtrace --reduce" in out


def test_gen_dataset_cli_round_trips_through_the_parser(tmp_path, built):
    """python -m aim_amd.gen_dataset writes the reference input format; parsing it back (get_reads semantics) yields the
    generator's pairs; chunking does not change the data; the oracle CLI aligns it."""
    import subprocess, sys
    import numpy as np
    from aim_amd import engine
    from conftest import ROOT
    f1, f2 = tmp_path / "a.seq", tmp_path / "b.seq"
    for f, chunk in ((f1, "1000"), (f2, "7")):
        subprocess.run([sys.executable, "-m", "aim_amd.gen_dataset", "-n", "50", "-l", "100", "-e", "0.02", "-s", "7", "-o", str(f),
                        "--chunk", chunk], check=True, cwd=ROOT)
    assert f1.read_bytes() == f2.read_bytes()
    data = f1.read_bytes()
    assert data.count(b"\n") == 100 and data.startswith(b">")
    ms, rs = engine.launcher_sizes("wfa", 100, 0.02)
    req, pat, txt = engine.parse_pairs(data, rs)
    greq, gpat, gtxt = engine.gen_pairs(7, 0, 50, 100, 0.02, rs)
    assert np.array_equal(req["pattern_len"], greq["pattern_len"]) and np.array_equal(req["text_len"], greq["text_len"])
    assert np.array_equal(pat, gpat) and np.array_equal(txt, gtxt)
    assert (req["pattern_len"] == 100).all() and (np.abs(req["text_len"].astype(int) - 100) <= 2).all()
