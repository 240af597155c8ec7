"""dp_strip_shape's cost rule against the measurements it was derived from (round 6, NOTES R6.7): for every row of the committed sweeps
(profiles/r06/strip_shape_sweep.txt: 1 024 pairs; strip_shape_sweep_few.txt: 256 / 512 pairs -- tools/strip_shape_sweep.py forces 16 / 20 / 24 / 32 cells per
lane on one MI355X) the shape the PLAN chooses for that READ_SIZE / batch size must be the fastest measured one or within 20 % of it. Runs on the CPU: planning is
pure (aim_plan_describe), the numbers are fixtures."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROW = re.compile(r"^(nw|swg)\s+(score|cigar) rs=(\d+) n=(\d+)\s+\|(.*)$")
CELL = re.compile(r"(\d+): dp_strip w(\d+) k(\d+) ([\d.]+) ms (\d+)")


def _rows(name):
    for line in open(os.path.join(ROOT, "profiles", "r06", name)):
        m = ROW.match(line)
        if not m:
            continue
        forced = {int(k): (int(w), float(g)) for k, w, k2, ms, g in CELL.findall(m.group(5)) if int(k) == int(k2)}
        if len(forced) == 4:
            yield m.group(1), m.group(2) == "cigar", int(m.group(3)), int(m.group(4)), forced


@pytest.mark.parametrize("sweep", ["strip_shape_sweep.txt", "strip_shape_sweep_few.txt"])
def test_plan_picks_a_measured_fast_shape(sweep, monkeypatch):
    from aim_amd import capi, engine
    lib = capi.load()
    for k in list(os.environ):
        if k.startswith("AIM_") and k != "AIM_LIB":
            monkeypatch.delenv(k)
    monkeypatch.setenv("AIM_NO_DP_GROUP", "1")          # the sweep's forced columns are dp_strip's; NW with CIGAR <= 2560 is dp_group's by default
    monkeypatch.setenv("AIM_CHIP_CUS", "256")
    rows = list(_rows(sweep))
    assert len(rows) >= 20
    misses = []
    for algo, bt, rs, n, forced in rows:
        params = engine.make_params(algo, 400, rs, backtrace=bt, swg_w16=(algo == "swg"))
        buf = C.create_string_buffer(1024)
        assert lib.aim_plan_describe(C.byref(params), n, buf, 1024) == 0
        line = buf.value.decode()
        assert line.startswith("dp_strip_kernel"), line
        k = int(re.search(r"cells_per_lane=(\d+)", line).group(1))
        w = int(re.search(r"wavefronts_per_pair=(\d+)", line).group(1))
        assert forced[k][0] == w, (line, forced)
        best = max(g for _, g in forced.values())
        if forced[k][1] < 0.80 * best:
            misses.append((algo, bt, rs, n, k, forced))
    assert not misses, misses
