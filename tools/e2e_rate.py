#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate through the C-ABI set calls: pinned host buffers -> aim_set_push ->
aim_set_launch -> aim_set_pull, i.e. AIM's CPU-DPU + DPU Kernel + DPU-CPU timers (host.c:270,297,328).
Reported next to the HBM-resident figure of bench.py; never used as bench.py's `value`.

    python tools/e2e_rate.py [--pairs 4194304] [--backtrace]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine  # noqa: E402


def pinned(lib, nbytes):
    p = C.c_void_p()
    capi.check(lib.aim_host_alloc(C.byref(p), nbytes))
    return p, np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p.value))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1 << 22)
    ap.add_argument("--backtrace", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    lib = capi.load()
    n = a.pairs
    ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
    params = engine.make_params("wfa", ms, rs, reduce=True, backtrace=a.backtrace)
    req, pat, txt = engine.gen_pairs(42, 0, n, 100, 0.01, rs)
    bufs = []
    for arr in (req, pat, txt):
        p, view = pinned(lib, arr.nbytes)
        view[:] = arr.view(np.uint8).reshape(-1)
        bufs.append((p, view))
    pres, _ = pinned(lib, n * capi.RESULT_DTYPE.itemsize)
    pops = pinned(lib, n * 2 * rs)[0] if a.backtrace else None
    s = C.c_void_p()
    capi.check(lib.aim_set_alloc(1, None, C.byref(s)))
    capi.check(lib.aim_set_configure(s, C.byref(params), n))
    wall = []
    for _ in range(a.reps + 1):
        t0 = time.perf_counter()
        capi.check(lib.aim_set_push(s, 0, n, bufs[0][0], bufs[1][0], bufs[2][0]))
        capi.check(lib.aim_set_launch(s))
        capi.check(lib.aim_set_pull(s, 0, pres, pops))
        wall.append(time.perf_counter() - t0)
    h2d, k, d2h = C.c_float(), C.c_float(), C.c_float()
    lib.aim_set_timers(s, C.byref(h2d), C.byref(k), C.byref(d2h))
    reps = a.reps + 1
    best = min(wall[1:])
    print(json.dumps({"pairs": n, "backtrace": a.backtrace, "e2e_pairs_per_s": n / best, "wall_ms_best": best * 1e3,
                      "h2d_ms": h2d.value / reps, "kernel_ms": k.value / reps, "d2h_ms": d2h.value / reps,
                      "h2d_GBps": (req.nbytes + pat.nbytes + txt.nbytes) / (h2d.value / reps * 1e-3) / 1e9}))
    lib.aim_set_free(s)


if __name__ == "__main__":
    main()
