"""MAX_SCORE / READ_SIZE heuristics against goldens produced by running the reference launchers
(tests/golden/make_launcher_goldens.py)."""
import json
import os

import pytest

from conftest import GOLDEN


def _records():
    recs = json.load(open(os.path.join(GOLDEN, "launcher_goldens.json")))
    return [r for r in recs if r["make_line"]]


def test_launcher_sizes_match_reference(built):
    from aim_amd import engine
    from oracle import oracle
    recs = _records()
    assert len(recs) > 100
    for r in recs:
        kw = {}
        if r["costs"] is not None:
            m, x, g, a = r["costs"]
            kw = dict(mismatch=x, gap_o=g, gap_e=a, gap=g)
        got = engine.launcher_sizes(r["algo"], r["l"], r["e"], **kw)
        assert got == (r["MAX_SCORE"], r["READ_SIZE"]), r["make_line"]
        assert oracle.launcher_sizes(r["algo"], r["l"], r["e"], **kw) == got


def test_launcher_cli_flag_line(built):
    """aim_amd.launch renders the same -D flag set the reference launcher echoes (minus UPMEM-only sizing)."""
    from aim_amd import launch
    for r in _records():
        argv = ["-i", "in", "-l", str(r["l"]), "-e", repr(r["e"]), "-n", "100000", "-d", "4"]
        if r["costs"] is not None:
            m, x, g, a = r["costs"]
            argv += ["-m", str(m), "-x", str(x), "-g", str(g)]
            if r["algo"] != "nw":
                argv += ["-a", str(a)]
        if r["backtrace"]:
            argv.append("-b")
        if r["reduce"]:
            argv.append("-r")
        cfg = launch.parse(r["algo"], argv)
        flags = launch.flag_line(cfg)
        want = [t for t in r["make_line"].split("FLAGS=")[1].split() if not t.startswith("-DWRAM_SEGMENT")]
        assert flags.split() == want


def test_launcher_rejects_bad_costs(built):
    from aim_amd import launch
    for algo in ("wfa", "swg", "nw"):
        with pytest.raises(SystemExit):
            launch.parse(algo, ["-i", "in", "-l", "100", "-e", "0.01", "-n", "10", "-m", "1"])
        with pytest.raises(SystemExit):
            launch.parse(algo, ["-i", "in", "-l", "100", "-e", "0.01", "-n", "10", "-x", "0"])
