"""The C-ABI library on a box without a GPU: it loads, exports every symbol include/aim_hip.h declares, and every
compute entry point fails loudly (no CPU fallback).  No compute calls are made here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "aim_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(aim_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound(built):
    from aim_amd import capi
    lib = capi.load()
    declared = _declared_functions()
    assert len(declared) >= 18
    nm = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (aim_[a-z0-9_]+)", nm))
    for name in declared:
        assert name in exported, name + " declared in aim_hip.h but not exported"
        assert hasattr(lib, name)
        assert name in capi.SYMBOLS, name + " missing from the ctypes binding"
    assert exported <= set(declared), "exported but undeclared: %s" % (exported - set(declared))
    assert lib.aim_abi_version() == 2


def test_kernels_are_compiled_for_gfx950(built):
    from aim_amd import capi
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    blob = open(capi.LIB_PATH, "rb").read()
    for k in (b"wfa_lane_kernel", b"wfa_wave_kernel", b"nw_lane_kernel", b"swg_lane_kernel", b"nw_reg_kernel", b"swg_reg_kernel", b"genasm_wave_kernel"):
        assert k in blob


def test_no_device_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from aim_amd import capi, engine
    lib = capi.load()
    n = C.c_int(-1)
    assert lib.aim_device_count(C.byref(n)) == capi.AIM_ENODEV and n.value == 0
    assert b"no HIP device" in lib.aim_last_error()
    with pytest.raises(capi.AimError) as e:
        engine.DeviceSet(1)
    assert e.value.code == capi.AIM_ENODEV
    p = engine.make_params("wfa", 5, 112)
    rc = lib.aim_align_device(C.byref(p), 1, None, None, None, None, None, None, 0, None)
    assert rc == capi.AIM_ENODEV


def test_product_never_references_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "aim_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".c")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "aim_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
    ldd = subprocess.run(["ldd", os.path.join(ROOT, "aim_amd", "libaim_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_scratch_planning_is_pure(built):
    from aim_amd import capi, engine
    lib = capi.load()
    for algo, ms, rs, kw in (("wfa", 5, 112, {}), ("wfa", 250, 1064, dict(backtrace=True, reduce=True)), ("nw", 4, 112, {}),
                             ("swg", 5, 112, dict(backtrace=True))):
        p = engine.make_params(algo, ms, rs, **kw)
        a = lib.aim_scratch_bytes(C.byref(p), 1 << 20)
        assert a > 0 and a == lib.aim_scratch_bytes(C.byref(p), 1 << 20)
    assert lib.aim_scratch_bytes(C.byref(engine.make_params("wfa", 5, 110)), 16) == 0   # invalid read_size
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 5, 112))) == b"wfa_lane_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 5, 112, backtrace=True))) == b"wfa_lane_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 5, 112, backtrace=True, mismatch=4))) == b"wfa_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 10, 112))) == b"wfa_lane_kernel"          # round 2: dynamic-bounds shape, score-only
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 10, 112, backtrace=True))) == b"wfa_lane_packed_kernel"   # round 3: CIGAR at MAX_SCORE 6..10 (rows packed on the device)
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 8, 160))) == b"wfa_lane_packed_kernel"                    # round 3: READ_SIZE beyond 80 / 112 (l = 150)
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 11, 112, backtrace=True))) == b"wfa_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 11, 112))) == b"wfa_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 250, 1064, backtrace=True, reduce=True))) == b"wfa_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 2000, 8000))) == b"wfa_wave_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 500, 10112, backtrace=True))) == b"dp_strip_kernel"    # round 3: column-strip pipeline
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 100, 1064, backtrace=True))) == b"swg_lane_kernel"     # int8 cells (MAX_SCORE < 127): the literal kernels -- one pair per lane while 64 lanes' rows fit LDS (round 6) ...
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 100, 1200, backtrace=True))) == b"dp_wave_kernel"      # ... beyond READ_SIZE 1199 the row-scan kernel's literal path
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 400, 20000))) == b"dp_wave_kernel"                      # beyond 16 384 columns
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 112))) == b"nw_reg_kernel"                           # round 4: rows in registers up to READ_SIZE 128 (112 with CIGAR)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 120, backtrace=True))) == b"nw_reg_kernel"          # round 5: READ_SIZE 120 / 128 with CIGAR too (l = 100, e = 10 %)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 160, backtrace=True))) == b"nw_reg_kernel"         # (l = 150: the pattern row in LDS, 12 dwords of direction bits)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 184))) == b"dp_group_kernel"                         # round 5: medium reads, G lanes per pair (READ_SIZE 177 .. 1024)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1024, backtrace=True))) == b"dp_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1032))) == b"dp_group_kernel"                       # round 6: score-only to READ_SIZE 1792 (NW; 20 / 24 / 28 registers per lane) ...
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1792))) == b"dp_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1800))) == b"dp_strip_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 1536))) == b"dp_group_kernel"                     # ... / 1536 (SWG: 16 / 20 / 24 registers)
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 1544))) == b"dp_strip_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1032, backtrace=True))) == b"dp_group_kernel"       # ... with CIGAR NW to READ_SIZE 1280 (20 registers per lane, 16-byte lane words),
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1288, backtrace=True))) == b"dp_strip_kernel"       #     READ_SIZE 1281 .. 1439 (SWG: 1025 .. 1439) stay on the strips,
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 1032, backtrace=True))) == b"dp_strip_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 40, 1440, backtrace=True))) == b"dp_group_kernel"       #     1440 .. 2048 are one pair of 45 .. 64 lanes per wavefront
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 2048, backtrace=True))) == b"dp_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 2056, backtrace=True))) == b"dp_strip_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 336, backtrace=True))) == b"dp_group_kernel"      # (int16 cells by MAX_SCORE)
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 8, 336, swg_w16=True))) == b"dp_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 8, 336))) == b"swg_lane_kernel"                        # (int8 cells wrap by design: the literal kernels)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 184, gap=180))) == b"nw_lane_kernel"                 # (dp_strip_exact_ok: an int16 store could wrap)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 184, gap=90))) == b"dp_group_kernel"                 # (round 6: READ_SIZE x gap bounds a cell, not twice that)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 300, 7904))) == b"dp_strip_kernel"                      # (until round 6 the literal one-lane path from READ_SIZE 3 998 on)
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 300, 8000))) == b"dp_wave_kernel"                       # (cells of unrelated sequences can wrap: literal)
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 4, 112))) == b"swg_reg_kernel"                          # round 5: M and I rows in registers up to READ_SIZE 128
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 200, 112, backtrace=True))) == b"swg_reg_kernel"       # (int16 cells too)
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 8, 160, swg_w16=True))) == b"swg_reg_kernel"            # (l = 150, int16 cells: the pattern row in LDS, M and I in 154 registers)
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 8, 160))) == b"swg_lane_kernel"                         # (l = 150, int8 cells: o + v e passes 127 -- every pair wraps (S3): no point in the register pass)
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 4, 184))) == b"swg_lane_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 4, 112, match=-1))) == b"swg_lane_kernel"              # (negative costs: cells may be negative without a wrap)


def _plan_line(params, n, env):
    """The [aim plan] line a fresh process prints for these parameters (the budget default is cached per process)."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from aim_amd import capi, engine; lib = capi.load(); "
            "p = engine.make_params(%r, %d, %d, reduce=True, backtrace=%r); print('SCRATCH', lib.aim_scratch_bytes(C.byref(p), %d))"
            % (ROOT, "wfa" if params[0] == "wfa" else params[0], params[1], params[2], params[3], n))
    e = dict(os.environ, AIM_PLAN_DEBUG="1", **env)
    r = subprocess.run([os.sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    plan = [l for l in r.stderr.splitlines() if l.startswith("[aim plan] wfa_group G=")]
    scratch = int(re.search(r"SCRATCH (\d+)", r.stdout).group(1))
    return (plan[-1] if plan else ""), scratch


def test_group_plan_prefers_a_wavefront_per_pair_when_lds_starves_residency(built):
    """wfa_group_plan (measured rule): a G <= 16 plan that LDS holds below 6 workgroups per CU yields to
    G = 64; plans with enough residency keep G <= 16; AIM_GROUP_G forces either."""
    from aim_amd import engine
    def G(l, e, **env):
        ms, rs = engine.launcher_sizes("wfa", l, e)
        line, _ = _plan_line(("wfa", ms, rs, False), 1 << 16, env)
        m = re.search(r"G=(\d+) ring_m=(\d+) .* per_cu=(\d+)", line)
        assert m, line
        return int(m.group(1)), int(m.group(2)), int(m.group(3))
    g, ring_m, per_cu = G(1000, 0.05)            # config 3: 503 diagonals, but WFA-adaptive keeps ~23 alive: narrow 128-entry rows,
    assert (g, ring_m) == (16, 6) and per_cu >= 12     # FOUR pairs per wavefront (round 3: the step is VALU-bound once the traceback is a kernel of its own); exact-size ring: max(x, o+e) + 1 = 6 rows
    g, ring_m, per_cu = G(1000, 0.05, AIM_GROUP_WLDS="0")   # one home per diagonal (round-1 layout, 10.7 KB per pair): 7 per CU
    assert (g, ring_m) == (32, 6) and per_cu == 7
    assert G(1000, 0.05, AIM_GROUP_WLDS="0", AIM_GROUP_G="64")[2] == 14
    assert G(400, 0.10)[0] == 16 and G(400, 0.10, AIM_GROUP_WLDS="0")[0] == 32   # narrow rows: 4 pairs per wavefront; one home per diagonal: 2
    assert G(250, 0.10)[0] in (16, 32) and G(250, 0.10, AIM_GROUP_WLDS="0")[0] == 32
    assert G(100, 0.10)[0] == 16 and G(100, 0.02, AIM_NO_LANE_EXT="1")[0] <= 16   # (e = 2 % score-only now runs on wfa_lane_kernel)
    assert G(1000, 0.05, AIM_GROUP_G="16")[0] == 16 and G(100, 0.10, AIM_GROUP_G="64")[0] == 64 and G(1000, 0.05, AIM_GROUP_G="64")[0] == 64
    # residency comes from the 1280-B LDS granule (aim_device.hpp: lds_workgroups_per_cu), capped at 16: a byte-granular
    # estimate would say 12 for the 12.8-KB and 13.3-KB workgroups, which measurably breaks into two rounds
    assert G(100, 0.02, AIM_NO_LANE_EXT="1")[2] == 11          # G = 4: 12.8 KB incl. LDS staging rows
    # round 2: groups of >= 8 lanes pack straight from global memory (no staging rows) and up to 20 workgroups per CU are used
    assert G(250, 0.05)[2] == 14 and G(100, 0.05)[2] == 16 and G(100, 0.10)[2] == 16   # 11.2 KB / 9.4 KB / 8.8 KB workgroups (18 fit: kept a multiple of the 4 SIMDs)
    assert G(1000, 0.05, AIM_GROUP_WLDS="128")[2] == 12    # score-only, 12.7-KB workgroups of four pairs with rows of 128: 12 fit
    assert G(1000, 0.05)[2] == 16 and G(400, 0.10)[2] == 14  # rows of 96 where 4 * MAX_SCORE <= READ_SIZE (10.1-KB workgroups); 128 for wider wavefronts
    assert G(1000, 0.05, AIM_GROUP_WLDS="96", AIM_GROUP_G="32")[0] == 32   # (only the G = 16 kernels address rows that are no power of two: back to 128)
    assert G(1000, 0.05, AIM_GROUP_WLDS="0", AIM_GROUP_G="64")[2] == 14              # 10.7 KB


def test_dp_group_plan_scratch_and_lds(built):
    """dp_group_kernel's plan (dp_group.hpp, round 5): with CIGAR every pair of a wavefront owns a slab of [READ_SIZE + 3][G] lane words (SWG 16 bytes, NW 8) + a byte per boundary cell;
    the to-do region sits behind the larger of (slabs, the fallback kernel's scratch); LDS = the pairs' slots + the traceback's 3-KB window."""
    import ctypes as C
    from aim_amd import capi, engine
    lib = capi.load()
    env = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("AIM_") and k != "AIM_LIB"}
    try:
        def shape(rs, bt):   # dp_group_kp (round 6): the registers per lane (NW: 16 / 20 / 24 / 28; with CIGAR 16 / 20) with the most pairs x resident wavefronts per register
            slot = 2 * ((rs + 79) & ~15) + 4 * ((rs + 47) & ~7) + 16
            best = None
            for kp in range(16, (20 if bt else 28) + 1, 4):
                g = (rs + 2 * kp - 1) // (2 * kp)
                p = 64 // g
                lds = ((p * slot + 15) & ~15) + (64 * 3 * 16 if bt else 0) + 64
                waves = min(8, max(1, 160 * 1024 // ((lds + 1279) // 1280 * 1280)))
                if best is None or p * waves * best[0] > best[1] * kp:
                    best = (kp, p * waves, g, p)
            return best[0], best[2], best[3], slot
        for rs in (192, 336, 736, 1024):
            for bt in (False, True):
                kp, G, P, slot = shape(rs, bt)
                p = engine.make_params("nw", 40, rs, backtrace=bt)
                buf = C.create_string_buffer(1024)
                assert lib.aim_plan_describe(C.byref(p), 1 << 16, buf, 1024) == 0
                line = buf.value.decode()
                assert line.startswith("dp_group_kernel") and "lanes_per_pair=%d " % G in line, line
                lds = int(re.search(r"lds=(\d+)", line).group(1)); grid = int(re.search(r"grid=(\d+)", line).group(1))
                assert lds == ((P * slot + 15) & ~15) + (64 * 3 * 16 if bt else 0) + 64, (rs, bt, lds)
                assert grid == min(2048, ((((1 << 16) + P - 1) // P + 7) // 8) * 8)
                total = lib.aim_scratch_bytes(C.byref(p), 1 << 16)
                Gbt = shape(rs, True)[1]
                slab = (((rs + 3) * Gbt * (8 if shape(rs, True)[0] == 16 else 16) + (rs + 3) + 64) + 255) & ~255      # (NW: 8-byte lane words at 32 columns since round 6 -- eight registers per dword --, 16 at 40; SWG: 16)
                todo = ((16 + (1 << 16)) * 4 + 255) & ~255
                assert total >= (grid * P * slab if bt else 0) + todo and total < (1 << 33), (rs, bt, total)
    finally:
        os.environ.update(env)


def test_scratch_bound_default_and_override(built):
    """scratch_budget_bytes: without a device the planning queries use 16 GB; AIM_SCRATCH_GB overrides; plans are
    need-capped (the headline's scratch does not depend on the bound), config 4's table slabs scale with it."""
    from aim_amd import engine
    ms4, rs4 = engine.launcher_sizes("swg", 10000, 0.01)
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from aim_amd import capi, engine; lib = capi.load(); "
            "p4 = engine.make_params('swg', %d, %d, backtrace=True); ms, rs = engine.launcher_sizes('wfa', 100, 0.01); "
            "p2 = engine.make_params('wfa', ms, rs, reduce=True); "
            "print(lib.aim_scratch_bytes(C.byref(p4), 128), lib.aim_scratch_bytes(C.byref(p2), 1 << 22))" % (ROOT, ms4, rs4))
    def q(**env):
        e = {k: v for k, v in os.environ.items() if k != "AIM_SCRATCH_GB"}
        e.update(env)
        r = subprocess.run([os.sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        return tuple(int(x) for x in r.stdout.split())
    # round 5: a resident SWG pair's slab is FOUR DIRECTION BITS per cell (K = 20 cells per lane: 16 bytes per lane and row) + a byte per row -- 82 MB instead of
    # the 614 MB of three int16 planes; round 6: nothing else (round 5's pool of int16 tables for the literal path is gone with that path)
    per_pair = (((rs4 + 3) * (rs4 // 20 + 2) * 16 + rs4 + 3 + 64) + 255) & ~255
    c4_default, c2_default = q()
    c4_16, c2_16 = q(AIM_SCRATCH_GB="16")
    c4_100, c2_100 = q(AIM_SCRATCH_GB="100")
    c4_8, _ = q(AIM_SCRATCH_GB="8")
    assert c4_default == c4_16                               # no GPU here: 16 GB fallback
    assert c2_default == c2_16 == c2_100                     # need-capped plan
    assert c4_16 == 128 * per_pair and c4_16 <= 16 << 30     # one round of 128 pairs (round 4: five)
    assert c4_100 == c4_16                                   # need-capped too
    assert c4_8 <= 8 << 30 and c4_8 < c4_16                   # a smaller bound trims the grid: g slabs, g a multiple of 8
    assert c4_8 % per_pair == 0 and (c4_8 // per_pair) % 8 == 0


def test_plans_follow_the_device_compute_unit_count(built):
    """Every persistent grid is `workgroups per CU x the device's CU count` (aim::resident_grid), not a literal 256: plans made for a
    64-CU (QPX partition), 128-CU (DPX) and 304-CU device scale with it, stay multiples of 8 (xcd_unit), and keep the per-CU residency."""
    from aim_amd import engine
    cases = [("wfa", 100, 0.01, {}, 1 << 22), ("wfa", 100, 0.05, {}, 1 << 22), ("wfa", 1000, 0.05, dict(reduce=True, backtrace=True), 1 << 16),
             ("wfa", 10000, 0.01, dict(reduce=True), 1 << 16), ("nw", 100, 0.01, dict(backtrace=True), 1 << 22), ("swg", 100, 0.01, {}, 1 << 22),
             ("swg", 1000, 0.05, {}, 1 << 16), ("genasm", 1000, 0.1, {}, 1 << 16)]
    lines = ["import ctypes as C, sys; sys.path.insert(0, %r); from aim_amd import capi, engine; lib = capi.load(); buf = C.create_string_buffer(512)" % ROOT]
    for algo, l, e, kw, n in cases:
        ms, rs = engine.launcher_sizes("wfa" if algo == "genasm" else algo, l, e)
        lines.append("p = engine.make_params(%r, %d, %d, **%r); assert lib.aim_plan_describe(C.byref(p), %d, buf, 512) == 0; print(buf.value.decode())"
                     % (algo, ms, rs, kw, n))
    def grids(cus):
        e = {k: v for k, v in os.environ.items() if k != "AIM_CHIP_CUS"}
        if cus:
            e["AIM_CHIP_CUS"] = str(cus)
        r = subprocess.run([os.sys.executable, "-c", "\n".join(lines)], env=e, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        out = [(l.split()[0], int(re.search(r"grid=(\d+)", l).group(1))) for l in r.stdout.splitlines()]
        assert len(out) == len(cases)
        return out
    whole = grids(256)
    assert grids(None) == whole                       # no device here: the whole chip
    assert dict(whole)["wfa_lane_kernel"] == 2048     # the headline plan is unchanged
    for cus in (64, 128, 304):
        for (k0, g0), (k1, g1) in zip(whole, grids(cus)):
            assert k0 == k1 and g1 % 8 == 0
            exp = g0 * cus // 256
            assert g1 == exp or (k0.startswith("dp_") and exp - 64 <= g1 <= exp), (k0, cus, g0, g1)   # (table slabs: the 16-GB planning bound may trim a few workgroups)


def test_round4_plan_shapes(built):
    """The shapes round 4 chose, as plans (no device needed): the strip kernel's 20 cells per lane -- 8 wavefronts = two per SIMD on every SIMD for
    READ_SIZE ~10 000 with about a pair per CU, one wavefront per pair at READ_SIZE 1 064 --, GenASM's single kernel at 24 wavefronts per CU
    with 4 KB of LDS, NW on short reads with the row in registers up to READ_SIZE 128 (112 with CIGAR)."""
    import ctypes as C
    from aim_amd import capi, engine
    lib = capi.load()

    def plan(p, n):
        buf = C.create_string_buffer(1024)
        assert lib.aim_plan_describe(C.byref(p), n, buf, 1024) == 0
        return buf.value.decode()
    env = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("AIM_") and k != "AIM_LIB"}
    try:
        l = plan(engine.make_params("swg", 500, 10112, backtrace=True), 256)
        assert "dp_strip_kernel" in l and "cells_per_lane=20" in l and "wavefronts_per_pair=8" in l and "block=512" in l
        l = plan(engine.make_params("swg", 250, 1064, backtrace=True), 65536)
        assert "cells_per_lane=20" in l and "wavefronts_per_pair=1" in l
        l = plan(engine.make_params("swg", 150, 3064, backtrace=True), 1024)
        assert "cells_per_lane=24" in l and "wavefronts_per_pair=2" in l                                      # (round 6: until then 16 x 3 -- 518 against 1 304 GCUPS at READ_SIZE 2 952, profiles/r06/strip_shape_sweep.txt)
        l = plan(engine.make_params("nw", 400, 2112), 1024)
        assert "cells_per_lane=20" in l and "wavefronts_per_pair=2" in l
        l = plan(engine.make_params("nw", 400, 2112, backtrace=True), 1024)
        assert l.startswith("dp_group_kernel") and "lanes_per_pair=53" in l, l                                # (NW with CIGAR: 40 columns per lane to READ_SIZE 2 560)
        l = plan(engine.make_params("nw", 700, 3688, backtrace=True), 4096)
        assert "cells_per_lane=32" in l and "wavefronts_per_pair=2" in l
        for rs, n, grid in ((110008, 4096, 4096), (110008, 1 << 20, 24 * 256), (120, 1 << 18, 24 * 256)):
            l = plan(engine.make_params("genasm", 0, rs, backtrace=True), n)
            assert l.startswith("genasm_wave_kernel") and "grid=%d " % grid in l and "lds=4160" in l, l
        assert plan(engine.make_params("nw", 4, 128), 1 << 20).startswith("nw_reg_kernel")
        assert plan(engine.make_params("nw", 4, 128, backtrace=True), 1 << 20).startswith("nw_reg_kernel")     # round 5: 62 registers = the 8 dwords of direction bits a row has
        l = plan(engine.make_params("nw", 4, 184, backtrace=True), 1 << 20)                                   # round 5: dp_group_kernel, 6 lanes per pair = 10 pairs per wavefront; its to-do list goes to nw_lane_kernel
        assert l.startswith("dp_group_kernel") and "lanes_per_pair=6" in l and "grid=2048" in l and "fb_block=64" in l, l
        l = plan(engine.make_params("swg", 200, 736, backtrace=True), 1 << 16)                                # ... READ_SIZE > 320: to dp_strip_kernel in to-do mode
        assert l.startswith("dp_group_kernel") and "lanes_per_pair=23" in l and "fb_block=64" in l, l
        os.environ["AIM_NO_DP_GROUP"] = "1"
        assert plan(engine.make_params("nw", 4, 184, backtrace=True), 1 << 20).startswith("nw_lane_kernel")
        assert plan(engine.make_params("swg", 200, 736, backtrace=True), 1 << 16).startswith("dp_strip_kernel")
        os.environ.pop("AIM_NO_DP_GROUP")
        assert plan(engine.make_params("nw", 4, 48), 1 << 20).startswith("nw_reg_kernel") and plan(engine.make_params("swg", 4, 88), 1 << 20).startswith("swg_reg_kernel")
        assert plan(engine.make_params("nw", 4, 112, backtrace=True), 1 << 20).startswith("nw_reg_kernel")
        assert plan(engine.make_params("nw", 4, 112, gap=60), 1 << 20).startswith("nw_lane_kernel")     # costs too large for INF = 16 000 to stay out of reach
    finally:
        os.environ.update(env)


def test_int8_swg_medium_reads_keep_a_plan_under_a_small_scratch_bound(monkeypatch):
    """Round 6: SWG with int8 cells at READ_SIZE 321 .. 1199 runs on swg_lane_kernel (one pair per lane). With CIGAR a workgroup's 64 tables are READ_SIZE^2 x 256 bytes;
    a scratch bound that cannot hold the smallest grid of them must fall back to dp_wave_kernel's one table per workgroup, not fail (ADVICE r05 on dp_group, same rule)."""
    from aim_amd import capi, engine
    lib = capi.load()
    def plan(p, n):
        buf = C.create_string_buffer(1024)
        assert lib.aim_plan_describe(C.byref(p), n, buf, 1024) == 0, lib.aim_last_error()
        return buf.value.decode()
    for k in list(os.environ):
        if k.startswith("AIM_") and k != "AIM_LIB":
            monkeypatch.delenv(k)
    assert plan(engine.make_params("swg", 100, 1064, backtrace=True), 4096).startswith("swg_lane_kernel")
    monkeypatch.setenv("AIM_SCRATCH_GB", "1")
    assert plan(engine.make_params("swg", 100, 1064, backtrace=True), 4096).startswith("dp_wave_kernel")
    assert plan(engine.make_params("swg", 100, 1064), 4096).startswith("swg_lane_kernel")            # score-only: no table at all
    assert plan(engine.make_params("swg", 100, 536, backtrace=True), 4096).startswith("swg_lane_kernel")
