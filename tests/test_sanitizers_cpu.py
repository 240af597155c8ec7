"""Sanitizer runs of the CPU-side code: the drop-in host program `aim_amd/host/host.c` and the oracle CLI are
built with AddressSanitizer + UndefinedBehaviorSanitizer and driven through their parser / packer / partition / writer paths. HOST code only -- there is no
GPU sanitizer on this pool. The `-m gpu` leg runs the same ASan host binary through a whole multi-shard, small-batch run on the device."""
import hashlib
import os
import subprocess

import pytest

from conftest import ROOT, judge_case_input, judge_cases, md5

SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
# (use_sigaltstack=0: with the HIP runtime in the process ASan cannot unmap its per-thread alternate signal stacks at thread exit and aborts in AsanThread::Destroy --
#  its own bookkeeping, not a finding; leaks are not checked: the HIP runtime keeps its allocations until exit)
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:use_sigaltstack=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")


@pytest.fixture(scope="session")
def host_asan(built):
    """aim_amd/host/host.c with ASan + UBSan, linked against the in-tree libaim_hip.so like the product binary (build/host_asan: not shipped)."""
    out = os.path.join(ROOT, "build", "host_asan")
    src = os.path.join(ROOT, "aim_amd", "host", "host.c")
    lib = os.path.join(ROOT, "aim_amd", "libaim_hip.so")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu11", "-Wall"] + SAN + ["-I" + os.path.join(ROOT, "include"), "-o", out, src,
                               "-L" + os.path.join(ROOT, "aim_amd"), "-laim_hip", "-Wl,-rpath," + os.path.join(ROOT, "aim_amd"), "-lm", "-lpthread"])
    return out


@pytest.fixture(scope="session")
def oracle_cli_asan(built):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle_cli_asan"])
    return os.path.join(ROOT, "oracle", "oracle_cli_asan")


def _run(cmd, cwd):
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, cwd=str(cwd), env=ENV)
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr and r.returncode not in (97, 98), r.stderr[-4000:]
    return r


def test_host_parser_and_packer_under_asan_ubsan(host_asan, sample_bytes, tmp_path):
    """`host --pack-only` (mmap, the threaded line index, validation, SSSE3/BMI2 packer, raw side list, batch file writer; no GPU is touched) on the
    reference's sample file and on the inputs its parser treats specially: reads with 'N' (raw side list), a last line without newline, an over-length
    read (message, exit 0, nothing written: get_reads, host.c:119-123), n <= NR_DPUS (exit 1, host.c:191-194) and n not a multiple of 8 * NR_DPUS (H3)."""
    from aim_amd import engine
    inp = tmp_path / "sample.seq"
    inp.write_bytes(sample_bytes)
    for threads, batch in ((1, 0), (5, 3000), (8, 1 << 20)):
        extra = ["--batch", batch] if batch else []
        r = _run([host_asan, inp, tmp_path / "o", 20000, "--read-size", 112, "--threads", threads, "--pack-only", tmp_path / ("d%d" % threads)] + extra, tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
    ref = (tmp_path / "d1").read_bytes()
    assert ref[:8] and hashlib.md5(ref).hexdigest() == hashlib.md5((tmp_path / "d8").read_bytes()).hexdigest()
    # 'N' reads, no final newline
    req, pat, txt = engine.gen_pairs(77, 0, 5000, 100, 0.02, 112)
    for i in range(0, 5000, 37):
        (pat if i % 2 else txt)[i, i % 90] = ord("N")
    data = engine.pairs_to_text(req, pat, txt)
    dirty = tmp_path / "dirty.seq"
    dirty.write_bytes(data[:-1])
    r = _run([host_asan, dirty, tmp_path / "o", 5000, "--read-size", 112, "--threads", 3, "--nr-dpus", 4, "--pack-only", tmp_path / "dd"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    # n is not a cap: 100 reads asked of 4 DPUs -> 128 consumed
    r = _run([host_asan, dirty, tmp_path / "o", 100, "--read-size", 112, "--nr-dpus", 4, "--pack-only", tmp_path / "d100"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    # over-length read
    lines = data.split(b"\n")
    lines[2 * 4000] += b"ACGT" * 10
    (tmp_path / "long.seq").write_bytes(b"\n".join(lines))
    r = _run([host_asan, tmp_path / "long.seq", tmp_path / "ol", 5000, "--read-size", 112, "--pack-only", tmp_path / "dl"], tmp_path)
    assert r.returncode == 0 and "READ LENGTH less than length of the input reads" in r.stdout and not (tmp_path / "dl").exists()
    # n <= NR_DPUS, a negative count, a missing file
    assert _run([host_asan, dirty, tmp_path / "o", 4, "--nr-dpus", 4, "--pack-only", tmp_path / "x"], tmp_path).returncode == 1
    assert _run([host_asan, dirty, tmp_path / "o", -5], tmp_path).returncode == 1
    assert _run([host_asan, tmp_path / "nope.seq", tmp_path / "o", 100, "--pack-only", tmp_path / "x"], tmp_path).returncode != 0


@pytest.mark.parametrize("name", ["nw_l40_e5_bt", "tails_swg_l100_bt_int8", "wfa_l70_ms6_bt_red_x5g4a2"])
def test_oracle_cli_under_asan_ubsan(oracle_cli_asan, name, tmp_path):
    """Three judge-r05 cases (NW with CIGAR; SWG int8 with tail-heavy pairs: the flat table's aliased reads; WFA-adaptive with CIGAR at MAX_SCORE 6) through the
    oracle's CLI built with ASan + UBSan: same digests, no report."""
    case = [c for c in judge_cases() if c["name"] == name][0]
    inp, out = tmp_path / "in", tmp_path / "out"
    inp.write_bytes(judge_case_input(case))
    g = case["gen"]
    cmd = [oracle_cli_asan, case["algo"], "-i", inp, "-o", out, "-n", g["n"], "-l", g["l"], "-e", g["e"], "-d", 1, "-t", 2, "--max-score", case["max_score"],
           "--read-size", case["read_size"], "-x", case["mismatch"]]
    cmd += ["-g", case.get("gap_o", case.get("gap_i", 4)), "-a", case.get("gap_e", 1)] + (["-b"] if case["backtrace"] else []) + (["-r"] if case.get("reduce") else [])
    r = _run(cmd, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert md5(out.read_bytes()) == case["output_md5"]


@pytest.mark.gpu
def test_host_pipeline_under_asan_on_the_device(host_asan, sample_bytes, ref_digests, tmp_path):
    """The ASan + UBSan host binary through whole runs on the GPU: three output shards (three lanes: pack pools, device sets, format pools, writers) with
    batches of 3 000 pairs, WFA with CIGAR and NW with CIGAR through the ops-row path -- the host's threads, slots, pinned buffers and writers under the sanitizer
    (HOST code only; the device library is the product build). Digests as the reference's."""
    inp = tmp_path / "sample.seq"
    inp.write_bytes(sample_bytes)
    for key, flags in (("wfa_backtrace", ["--algo", "wfa", "--max-score", 5, "--backtrace"]),
                       ("nw_backtrace", ["--algo", "nw", "--max-score", 4, "--backtrace", "--no-pack", "--full-ops"])):
        out = tmp_path / ("out_" + key)
        r = _run([host_asan, inp, out, 20000, "--read-size", 112, "--nr-dpus", 4] + flags + ["--out-shards", 3, "--batch", 3000], tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        parts = sorted(p for p in os.listdir(tmp_path) if p.startswith(out.name + "."))
        data = b"".join((tmp_path / p).read_bytes() for p in parts) if parts else out.read_bytes()
        assert md5(data) == ref_digests[key], (key, parts)
