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
    assert lib.aim_abi_version() == 1


def test_kernels_are_compiled_for_gfx950(built):
    from aim_amd import capi
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    blob = open(capi.LIB_PATH, "rb").read()
    for k in (b"wfa_lane_kernel", b"wfa_wave_kernel", b"nw_lane_kernel", b"swg_lane_kernel"):
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
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 10, 112))) == b"wfa_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 250, 1064, backtrace=True, reduce=True))) == b"wfa_group_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("wfa", 2000, 8000))) == b"wfa_wave_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("swg", 500, 10112, backtrace=True))) == b"dp_wave_kernel"
    assert lib.aim_kernel_name(C.byref(engine.make_params("nw", 4, 112))) == b"nw_lane_kernel"
