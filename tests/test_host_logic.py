"""Host-side logic that needs no GPU: parser (get_reads), writer / edit_cigar_print, generator, host CLI argument
handling -- compared with the oracle's restatement where one exists."""
import os
import subprocess

import numpy as np
import pytest

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


def test_host_cli_output_to_a_pipe_is_sequential(built):
    """the writer must not pwrite() into a pipe (ESPIPE). Only the code path can be checked without a GPU: host.c stats
    the output and switches writer + fallback loop to in-order write() when it is no regular file."""
    src = open(os.path.join(ROOT, "aim_amd", "host", "host.c")).read()
    assert "S_ISREG" in src and "w->seq ? write(" in src and "f->seq ? write(" in src


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
    assert "--slots" not in out
    # GenASM long reads: four batches in flight per device (a pair that loses the diagonal outlasts its batch)
    assert launch.main(["genasm", "-i", "a", "-o", "b", "-l", "100000", "-e", "0.1", "-n", "64", "-b", "--dry-run"]) == 0
    out = capsys.readouterr().out
    assert "--algo genasm" in out and "--read-size 110008" in out and "--slots 4" in out


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


def test_pack_sequence_and_numpy_packer_agree(built):
    """aim_pack_sequence (C, used by the CLI's parser threads) and engine.pack_rows (numpy, used by tests/bench) implement
    the packed-input format of aim_hip.h: code (ascii>>1)&3, base i at bits 2(i%16) of dword i/16; a byte outside A/C/G/T
    inside the sequence makes the pair travel raw; bytes beyond the length never matter."""
    import ctypes as C
    import numpy as np
    from aim_amd import capi, engine
    lib = capi.load()
    rng = np.random.RandomState(5)
    for rs in (80, 112, 120, 1064):
        req, pat, txt = engine.gen_pairs(9, 0, 200, int(rs / 1.06) - 4, 0.05, rs)
        pat[3, 5] = ord("N"); pat[4, int(req["pattern_len"][4])] = ord("N")      # inside / just outside the sequence
        pat[5, 0] = ord("a"); pat[6, 2] = 0xC1; pat[7, 1] = 0
        packed, ok = engine.pack_rows(req, pat, "pattern_len")
        dw = engine.packed_row_dwords(rs)
        assert packed.shape == (200, dw)
        for i in range(200):
            row = np.zeros(dw, dtype=np.uint32)
            r = lib.aim_pack_sequence(capi.ptr(pat[i]), int(req["pattern_len"][i]), rs, capi.ptr(row))
            assert r == int(ok[i]), i
            if r:
                assert np.array_equal(row, packed[i]), i
        assert list(np.nonzero(~ok)[0]) == [3, 5, 6, 7]
        # unpack model: "ACTG"[code] reproduces the rows up to their lengths
        lut = np.frombuffer(b"ACTG", dtype=np.uint8)
        for i in (0, 1, 4, 199):
            codes = (packed[i][:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3
            seq = lut[codes.reshape(-1)][: req["pattern_len"][i]]
            assert np.array_equal(seq, pat[i, : req["pattern_len"][i]])


def test_cigar_format_runs_equals_edit_cigar_print(built):
    """aim_cigar_format_runs prints (length<<8 | op) runs exactly like aim_cigar_format prints the ops they encode
    (edit_cigar_print, host.c:69-89), also when a run arrives split in two."""
    import ctypes as C
    import numpy as np
    from aim_amd import capi, engine
    lib = capi.load()
    rng = np.random.RandomState(11)
    for _ in range(200):
        n = int(rng.randint(1, 400))
        ops = rng.choice(np.frombuffer(b"MMMMMMXID", dtype=np.uint8), size=n)
        want = engine.cigar_of(ops, 0, n)
        runs, start = [], 0
        for i in range(1, n + 1):
            if i == n or ops[i] != ops[start]:
                ln = i - start
                if ln > 3 and rng.rand() < 0.3:      # a split run must be merged by the formatter
                    k = int(rng.randint(1, ln))
                    runs += [(k << 8) | int(ops[start]), ((ln - k) << 8) | int(ops[start])]
                else:
                    runs.append((ln << 8) | int(ops[start]))
                start = i
        r = np.array(runs, dtype=np.uint32)
        buf = C.create_string_buffer(12 * len(r) + 16)
        k = capi.check(lib.aim_cigar_format_runs(capi.ptr(r), len(r), buf, len(buf)))
        assert buf.raw[:k] == want


def _read_pack_dump(path):
    import numpy as np
    from aim_amd import capi, engine
    raw = open(path, "rb").read()
    assert raw[:8] == b"AIMPK\0\0\1" and len(raw) >= 64          # packed batch file: 64-byte header (magic, version, READ_SIZE, request bytes, batch, total)
    version, frs, rqb, fbatch = np.frombuffer(raw, dtype=np.uint32, count=4, offset=8)
    total = int(np.frombuffer(raw, dtype=np.uint64, count=1, offset=24)[0])
    assert version == 1 and rqb == 8
    at, jobs = 64, []
    while at < len(raw):
        n, ascii_, n_raw, rs = np.frombuffer(raw, dtype=np.uint32, count=4, offset=at); at += 16
        n, n_raw, rs = int(n), int(n_raw), int(rs)
        dw = engine.packed_row_dwords(rs)
        req = np.frombuffer(raw, dtype=capi.REQUEST8_DTYPE, count=n, offset=at); at += 8 * n
        job = dict(n=n, ascii=bool(ascii_), req=req, rs=rs)
        if ascii_:
            job["pat"] = np.frombuffer(raw, dtype=np.uint8, count=n * rs, offset=at).reshape(n, rs); at += n * rs
            job["txt"] = np.frombuffer(raw, dtype=np.uint8, count=n * rs, offset=at).reshape(n, rs); at += n * rs
        else:
            job["pkP"] = np.frombuffer(raw, dtype=np.uint32, count=n * dw, offset=at).reshape(n, dw); at += 4 * n * dw
            job["pkT"] = np.frombuffer(raw, dtype=np.uint32, count=n * dw, offset=at).reshape(n, dw); at += 4 * n * dw
            job["raw_idx"] = np.frombuffer(raw, dtype=np.uint32, count=n_raw, offset=at); at += 4 * n_raw
            job["rawP"] = np.frombuffer(raw, dtype=np.uint8, count=n_raw * rs, offset=at).reshape(n_raw, rs); at += n_raw * rs
            job["rawT"] = np.frombuffer(raw, dtype=np.uint8, count=n_raw * rs, offset=at).reshape(n_raw, rs); at += n_raw * rs
        assert rs == frs and n <= fbatch
        jobs.append(job)
    assert sum(j["n"] for j in jobs) == total
    return jobs


@pytest.mark.parametrize("threads,dirty", [(1, 40), (5, 40), (8, 3000)])
def test_host_cli_packer_matches_numpy_packer(built, tmp_path, threads, dirty):
    """The C host's parser threads (SSSE3/BMI2 packer, raw side list assembled from per-thread ranges, ASCII fallback
    for unusually dirty batches) produce exactly the batch engine.pack_batch builds (--pack-only: no GPU is touched)."""
    import subprocess
    import numpy as np
    from aim_amd import engine
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    n, rs = 20000, 112
    req, pat, txt = engine.gen_pairs(123, 0, n, 100, 0.02, rs)
    rng = np.random.RandomState(dirty)
    for i in rng.choice(n, size=dirty, replace=False):
        (pat if i % 2 else txt)[i, int(rng.randint(0, 90))] = ord("N")
    pat[17, :16] = ord("G"); txt[17, :16] = ord("G")               # a legal all-G head (packs to 0xffffffff)
    inp = tmp_path / "in.seq"
    inp.write_bytes(engine.pairs_to_text(req, pat, txt))
    dump = tmp_path / "dump.bin"
    r = subprocess.run([host, str(inp), str(tmp_path / "out"), str(n), "--read-size", str(rs), "--threads", str(threads),
                        "--pack-only", str(dump)], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    jobs = _read_pack_dump(dump)
    assert sum(j["n"] for j in jobs) == n
    at = 0
    for j in jobs:
        lo, hi = at, at + j["n"]
        at = hi
        assert np.array_equal(j["req"]["pattern_len"], req["pattern_len"][lo:hi]) and np.array_equal(j["req"]["idx"], req["idx"][lo:hi])
        assert np.array_equal(j["req"]["text_len"], req["text_len"][lo:hi])
        if j["ascii"]:
            assert dirty > j["n"] // 16
            assert np.array_equal(j["pat"], pat[lo:hi]) and np.array_equal(j["txt"], txt[lo:hi])
            continue
        pp, pt, raw, rawp, rawt = engine.pack_batch(req[lo:hi], pat[lo:hi], txt[lo:hi])
        assert np.array_equal(j["raw_idx"], raw)
        assert np.array_equal(j["rawP"], rawp) and np.array_equal(j["rawT"], rawt)
        keep = np.ones(j["n"], dtype=bool); keep[raw] = False
        assert np.array_equal(j["pkP"][keep], pp[keep]) and np.array_equal(j["pkT"][keep], pt[keep])
    if dirty == 40:
        assert not any(j["ascii"] for j in jobs) and sum(len(j["raw_idx"]) for j in jobs) == 40
    else:
        assert any(j["ascii"] for j in jobs)


def test_host_cli_validates_whole_input_before_writing(built, tmp_path):
    """an over-length read anywhere in the consumed range ends the run like get_reads does (message, exit 0)
    BEFORE anything is launched or written -- the output file stays empty; a negative read count is 'Invalid nb of reads'."""
    import subprocess
    from aim_amd import engine
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    req, pat, txt = engine.gen_pairs(5, 0, 3000, 100, 0.01, 112)
    lines = engine.pairs_to_text(req, pat, txt).split(b"\n")
    lines[2 * 2900] = lines[2 * 2900] + b"A" * 40
    inp = tmp_path / "in.seq"
    inp.write_bytes(b"\n".join(lines))
    out = tmp_path / "out"
    r = subprocess.run([host, str(inp), str(out), "3000", "--read-size", "112", "--pack-only", str(tmp_path / "d")],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and "READ LENGTH less than length of the input reads" in r.stdout
    assert out.read_bytes() == b"" and not (tmp_path / "d").exists()
    r = subprocess.run([host, str(inp), str(out), "-5"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Invalid nb of reads" in r.stderr


def test_gen_dataset_packed_equals_host_pack_only(built, tmp_path):
    """The packed batch file has ONE definition: `python -m aim_amd.gen_dataset --packed` (numpy packer) and `host --pack-only`
    (C packer on the text form of the same pairs) write the same bytes."""
    import subprocess, sys
    host = os.path.join(ROOT, "aim_amd", "host", "host")
    n, l, e = 20000, 100, 0.02
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.check_call([sys.executable, "-m", "aim_amd.gen_dataset", "-n", str(n), "-l", str(l), "-e", str(e), "-s", "9", "-o", str(tmp_path / "t.seq")], env=env)
    subprocess.check_call([sys.executable, "-m", "aim_amd.gen_dataset", "-n", str(n), "-l", str(l), "-e", str(e), "-s", "9", "-o", str(tmp_path / "g.aimpk"),
                           "--packed", "--batch", str(n)], env=env)
    r = subprocess.run([host, str(tmp_path / "t.seq"), str(tmp_path / "o"), str(n), "--read-size", "112", "--threads", "3", "--pack-only", str(tmp_path / "h.aimpk")],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "g.aimpk").read_bytes() == (tmp_path / "h.aimpk").read_bytes()
