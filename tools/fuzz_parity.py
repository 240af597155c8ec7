#!/usr/bin/env python3
"""Randomised differential run: HIP path (through the C-ABI) against the CPU oracle, bit-exact, over random algorithms,
lengths, error rates, penalties, flags, pair counts and forced plans. The oracle is used here as the checker only.

    python tools/fuzz_parity.py [--seconds 120] [--seed 1] [--max-cells 4e8]

Prints one line per case and a summary; exits 1 on the first mismatch (the case's parameters are printed, so it can be
turned into a fixed test)."""
import argparse, json, os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from aim_amd import capi, engine
from oracle import oracle
import ctypes as C


def oracle_params(params, algo):
    bt = bool(params.flags & capi.FLAG_BACKTRACE)
    red = bool(params.flags & capi.FLAG_REDUCE)
    cellb = 2 if (params.flags & capi.FLAG_SWG_W16) else 0
    return oracle.params(algo, params.max_score, params.read_size, match=params.match, mismatch=params.mismatch,
                         gap_o=params.gap_o, gap_e=params.gap_e, gap_i=params.gap_i, gap_d=params.gap_d, backtrace=bt, reduce=red, swg_cell_bytes=cellb)


def compare(algo, params, req, pat, txt, allow_nomem=False):
    """None if bit-identical. With allow_nomem (a tiny scratch bound was forced) pairs the HIP path ended with
    AIM_PAIR_NOMEM (3: history pool overflow, the counterpart of the reference arena's 'out of memory' abort, which the
    oracle does not model) are excluded; their count is returned through compare.nomem."""
    res, ops = engine.align(params, req, pat, txt, check=False)
    ores, oops, _ = oracle.align_batch(oracle_params(params, algo), req["pattern_len"], req["text_len"], pat, txt, nthreads=64)
    keep = (res["status"] != 3) if allow_nomem else np.ones(len(res), dtype=bool)
    compare.nomem = int((~keep).sum())
    for f in ("score", "max_operations", "end_offset", "status"):
        bad = np.nonzero((res[f] != ores[f]) & keep)[0]
        if bad.size: return "%s differs at pair %d: hip %d oracle %d" % (f, bad[0], res[f][bad[0]], ores[f][bad[0]])
    if params.flags & capi.FLAG_BACKTRACE:
        ok = (res["status"] == 0) & keep
        bad = np.nonzero((res["begin_offset"] != ores["begin_offset"]) & ok)[0]
        if bad.size: return "begin_offset differs at pair %d" % bad[0]
        for i in np.nonzero(ok)[0]:
            b, e = int(res["begin_offset"][i]), int(res["end_offset"][i])
            if not np.array_equal(ops[i, b:e], oops[i, b:e]): return "ops differ at pair %d" % i
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=float, default=4e8, help="bound on n * l * l per case (oracle time)")
    ap.add_argument("--focus", choices=["all", "lane", "genasm", "wfa", "fused", "dp", "dplane", "dpgroup"], default="all",
                    help="'lane': stay inside the one-pair-per-lane kernels' eligibility window (wfa_lane_kernel, wfa_lane_packed_kernel); 'genasm': GenASM only; "
                         "'wfa': the general generator restricted to WFA (wfa_group / wfa_wave / wfa_lane); 'fused': WFA batches through aim_set_submit as PACKED "
                         "rows with the compact CIGAR back (one kernel per batch: wfa_lane_packed / wfa_group + traceback kernel), output text against the oracle's; "
                         "'dp': NW / SWG long reads (dp_strip_kernel with every cells-per-lane shape, dp_wave_kernel); 'dpgroup': NW / SWG medium reads (dp_group_kernel, its to-do list and both fallbacks); "
                         "'dplane': NW / SWG short reads (nw_reg_kernel + its to-do pass, nw_lane_kernel, swg_lane_kernel): every length relation, outliers, costs")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    lib = capi.load()
    t0, cases, kernels, skipped, nomem_pairs = time.time(), 0, {}, 0, 0
    while time.time() - t0 < a.seconds:
        if a.focus == "lane":
            # default penalties, MAX_SCORE 0..5, READ_SIZE 80 or 112, any pair count (partial last groups), non-ACGT bytes
            rs = rng.choice([80, 112, 112, 136, 144, 160, 176])      # 80 / 112: wfa_lane_kernel (ASCII rows); the others: packed on the device
            l = rng.randint(1, rs - 12)
            e = rng.choice([0.0, 0.01, 0.02, 0.03, 0.05, 0.08, 0.10])
            if l + int(np.ceil(l * e)) + 1 > rs: e = 0.01
            ms = rng.randint(0, 10)                       # 6..10: the dynamic-bounds shape (WFA-adaptive's reduction can fire; CIGAR through the LDS history)
            n = rng.choice([1, 63, 64, 65, 127, 1000, 4097, 20000])
            kw = dict(backtrace=rng.random() < 0.6, reduce=rng.random() < 0.7)
            params = engine.make_params("wfa", ms, rs, **kw)
            for k in list(os.environ):
                if k.startswith("AIM_") and k != "AIM_LIB" and not k.startswith("AIM_DEBUG_POISON"): os.environ.pop(k)
            req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
            for _ in range(rng.choice([0, 0, 1, 5])):
                pat[rng.randrange(n), rng.randrange(max(1, l))] = ord(rng.choice("Nn*acgt"))
            kn = lib.aim_kernel_name(C.byref(params)).decode()
            case = dict(algo="wfa", l=l, e=e, n=n, max_score=ms, read_size=rs, kernel=kn, **{k: int(v) for k, v in kw.items()})
            try:
                err = compare("wfa", params, req, pat, txt)
            except Exception as ex:
                err = "exception: %r" % (ex,)
            cases += 1
            kernels[kn] = kernels.get(kn, 0) + 1
            if err:
                print(json.dumps(dict(case, ok=False)), flush=True)
                print("MISMATCH:", err, flush=True)
                return 1
            continue
        if a.focus == "dplane":
            # short-read NW / SWG: READ_SIZE 40 .. 128, lengths anywhere inside it, per-pair length outliers (tails plen >= tlen + 2, short reads
            # left of nw_reg_kernel's window), penalties on both sides of nw_reg_supported(), with and without the register kernel
            algo = rng.choice(["nw", "nw", "nw", "swg"])
            rs = rng.choice([40, 48, 64, 72, 80, 88, 96, 104, 112, 112, 112, 120, 128, 136, 144, 160, 160, 176])   # (> 128: swg_reg with the pattern row in LDS; NW: nw_lane)
            l = rng.randint(max(1, rs - 40), rs - 8)
            e = rng.choice([0.0, 0.01, 0.02, 0.05, 0.10])
            if l + int(np.ceil(l * e)) + 1 > rs: e = 0.0
            cost = dict(mismatch=rng.randint(1, 9), gap=rng.choice([1, 2, 3, 4, 5, 9, 30, 60])) if (algo == "nw" and rng.random() < 0.5) else {}
            if algo == "nw" and rng.random() < 0.4:                     # GAP_I != GAP_D (nw.c:67-153; nw_reg's tilt treats them separately)
                cost = dict(mismatch=rng.randint(1, 9), gap_i=rng.choice([1, 2, 3, 4, 6, 7, 9, 30]), gap_d=rng.choice([1, 2, 3, 5, 6, 7, 9, 30]))
            if algo == "swg" and rng.random() < 0.5:                   # swg_reg_kernel on both sides of swg_reg_supported(): other costs, int16 cells, MAX_SCORE as the pseudo-infinity
                cost = dict(mismatch=rng.randint(1, 9), gap_o=rng.choice([1, 2, 4, 6, 9, 40]), gap_e=rng.choice([1, 1, 2, 3, 5]))
                if rng.random() < 0.1: cost["match"] = rng.choice([-2, -1, 0])
            n = rng.choice([1, 63, 64, 65, 1000, 4097, 9000])
            bt = rng.random() < 0.6
            ms = rng.randint(1, 60)
            if algo == "swg":
                ms = rng.choice([0, 1, 5, 10, 25, 50, 100, 126, 127, 200, 2000])
                if rng.random() < 0.3: cost["swg_w16"] = True
            params = engine.make_params(algo, ms, rs, backtrace=bt, **cost)
            for k in list(os.environ):
                if k.startswith("AIM_") and k != "AIM_LIB" and not k.startswith("AIM_DEBUG_POISON"): os.environ.pop(k)
            env = {}
            if rng.random() < 0.15: env["AIM_NO_NW_REG"] = "1"
            if rng.random() < 0.1: env["AIM_NO_SWG_REG"] = "1"
            if rng.random() < 0.2: env["AIM_NW_REG_PER_CU"] = rng.choice(["1", "3", "16"])
            if rng.random() < 0.2: env["AIM_DPL_PER_CU"] = rng.choice(["1", "3", "12"])
            if rng.random() < 0.15: env["AIM_CHIP_CUS"] = rng.choice(["1", "2"])     # a small resident grid: many groups of 64 pairs per wavefront (the queues' carry-over)
            os.environ.update(env)
            req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
            if algo == "swg" and n > 8 and rng.random() < 0.4:          # unrelated texts: cells climb to MAX_SCORE + min(h, v) e, int8 cells wrap, the walk may find no operation
                for i in range(0, n, rng.choice([2, 7, 50])):
                    k = int(req["text_len"][i])
                    txt[i, :k] = np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(k)), dtype=np.uint8)
            for _ in range(rng.choice([0, 3, 40])):                     # outliers: shorter patterns / texts (contents stay what they were)
                i = rng.randrange(n)
                if rng.random() < 0.5: req["pattern_len"][i] = rng.randint(0, int(req["pattern_len"][i]))
                else: req["text_len"][i] = rng.randint(0, int(req["text_len"][i]))
            if n > 3 and rng.random() < 0.3: pat[rng.randrange(n), rng.randrange(max(1, l // 2))] = ord("N")
            if rng.random() < 0.3:                                      # shifted texts: paths off the corner-to-corner diagonal (round 6: the register kernels keep direction bits for a band only)
                for i in range(0, n, rng.choice([1, 3, 17])):
                    pl, sh = int(req["pattern_len"][i]), rng.randint(1, max(1, l // 2))
                    if pl <= sh + 1 or pl > rs: continue
                    t = np.concatenate([pat[i, sh:pl], np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(sh)), dtype=np.uint8)])
                    if rng.random() < 0.5: t = np.concatenate([t[pl - sh:], t[:pl - sh]])   # ... or shifted the other way
                    txt[i, :] = 0
                    txt[i, :pl] = t
                    req["text_len"][i] = pl
            kn = lib.aim_kernel_name(C.byref(params)).decode()
            case = dict(algo=algo, l=l, e=e, n=n, max_score=ms, read_size=rs, kernel=kn, backtrace=int(bt), cost=cost, env=env)
            try:
                err = compare(algo, params, req, pat, txt)
            except Exception as ex:
                err = "exception: %r" % (ex,)
            cases += 1
            kernels[kn] = kernels.get(kn, 0) + 1
            if err:
                print(json.dumps(dict(case, ok=False)), flush=True)
                print("MISMATCH:", err, flush=True)
                return 1
            continue
        if a.focus == "dpgroup":
            # medium-read NW / SWG (dp_group_kernel: G lanes per pair, READ_SIZE 177 .. 1024 and its neighbours): lengths anywhere inside the row, tails of
            # every size (plen > tlen: the aliased boundary cells, the last row's tail cells), outliers for the to-do list (empty sequences, plen > 2 tlen) and
            # its two fallbacks (dp_lane kernels up to READ_SIZE 320, dp_strip in to-do mode above), penalties on both sides of dp_strip_exact_ok()
            algo = rng.choice(["nw", "swg"])
            rs = rng.choice([176, 184, 192, 200, 224, 256, 264, 288, 320, 328, 336, 384, 416, 512, 520, 640, 728, 736, 992, 1000, 1024, 1032,
                             1040, 1232, 1280, 1288, 1432, 1440, 1488, 1536, 1544, 1736, 1792, 1800, 2000, 2048, 2056, 2112, 2424, 2560, 2568])   # (round 6: dp_group_rs_ok's ranges and their neighbours)
            l = rng.randint(max(1, rs - rs // 3), rs - 8)
            e = rng.choice([0.0, 0.01, 0.02, 0.05, 0.10, 0.15])
            if l + int(np.ceil(l * e)) + 1 > rs: e = 0.0
            cost = {}
            if algo == "nw" and rng.random() < 0.5: cost = dict(mismatch=rng.randint(1, 9), gap=rng.choice([1, 2, 3, 4, 5, 9, 14, 30]))
            if algo == "nw" and rng.random() < 0.4: cost = dict(mismatch=rng.randint(1, 9), gap_i=rng.choice([1, 2, 3, 4, 6, 7, 9, 30]), gap_d=rng.choice([1, 2, 3, 5, 6, 7, 9, 30]))
            if algo == "swg" and rng.random() < 0.5:
                cost = dict(mismatch=rng.randint(1, 9), gap_o=rng.choice([1, 2, 4, 6, 9, 40]), gap_e=rng.choice([1, 1, 2, 3, 5]))
                if rng.random() < 0.1: cost["match"] = rng.choice([-2, -1, 0])
            n = int(min(rng.choice([1, 5, 6, 7, 63, 64, 65, 500, 2000]), max(1, a.max_cells // (l * l))))
            bt = rng.random() < 0.6
            ms = rng.choice([1, 20, 60, 200, 600])
            if algo == "swg":
                ms = rng.choice([0, 25, 100, 126, 127, 200, 500, 2000])
                if rng.random() < 0.7: cost["swg_w16"] = True
            params = engine.make_params(algo, ms, rs, backtrace=bt, **cost)
            for k in list(os.environ):
                if k.startswith("AIM_") and k != "AIM_LIB" and not k.startswith("AIM_DEBUG_POISON"): os.environ.pop(k)
            env = {}
            if rng.random() < 0.12: env["AIM_NO_DP_GROUP"] = "1"
            if rng.random() < 0.2: env["AIM_DPG_PER_CU"] = rng.choice(["1", "3", "12"])
            if rng.random() < 0.15: env["AIM_CHIP_CUS"] = rng.choice(["1", "2", "8"])     # a small resident grid: many units per wavefront (LDS slots, slabs and the window reused)
            if rng.random() < 0.1: env["AIM_SCRATCH_GB"] = rng.choice(["0.5", "2"])
            os.environ.update(env)
            req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
            for _ in range(rng.choice([0, 3, 40])):                     # length outliers (contents stay what they were): tails of any size, plen > 2 tlen, empty sequences
                i = rng.randrange(n)
                r = rng.random()
                if r < 0.4: req["text_len"][i] = rng.randint(0, int(req["text_len"][i]))
                elif r < 0.7: req["pattern_len"][i] = rng.randint(0, int(req["pattern_len"][i]))
                elif r < 0.85: req["text_len"][i] = max(1, int(req["pattern_len"][i]) - rng.randint(1, 70))
                elif r < 0.93: req["text_len"][i] = max(1, (int(req["pattern_len"][i]) + 1) // 2 + rng.randint(-1, 1))
                else: req["text_len"][i] = max(1, int(req["pattern_len"][i]) // rng.choice([3, 4, 7, 30, 2000]))   # plen > 2 tlen: the last row's tail wraps the flat table more than once
            if n > 8 and rng.random() < 0.3:                            # unrelated texts: every cell on the gap / mismatch branches
                for i in range(0, n, rng.choice([2, 7, 50])):
                    k = int(req["text_len"][i])
                    txt[i, :k] = np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(k)), dtype=np.uint8)
            if n > 3 and rng.random() < 0.3: pat[rng.randrange(n), rng.randrange(max(1, l // 2))] = ord("N")
            if rng.random() < 0.3:                                      # shifted texts: paths off the corner-to-corner diagonal (round 6: the register kernels keep direction bits for a band only)
                for i in range(0, n, rng.choice([1, 3, 17])):
                    pl, sh = int(req["pattern_len"][i]), rng.randint(1, max(1, l // 2))
                    if pl <= sh + 1 or pl > rs: continue
                    t = np.concatenate([pat[i, sh:pl], np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(sh)), dtype=np.uint8)])
                    if rng.random() < 0.5: t = np.concatenate([t[pl - sh:], t[:pl - sh]])   # ... or shifted the other way
                    txt[i, :] = 0
                    txt[i, :pl] = t
                    req["text_len"][i] = pl
            kn = lib.aim_kernel_name(C.byref(params)).decode()
            case = dict(algo=algo, l=l, e=e, n=n, max_score=ms, read_size=rs, kernel=kn, backtrace=int(bt), cost=cost, env=env)
            try:
                err = compare(algo, params, req, pat, txt, allow_nomem="AIM_SCRATCH_GB" in env)
            except Exception as ex:
                err = "exception: %r" % (ex,)
                if "AIM_SCRATCH_GB" in env and "error -3" in err:
                    skipped += 1
                    continue
            cases += 1
            kernels[kn] = kernels.get(kn, 0) + 1
            if err:
                print(json.dumps(dict(case, ok=False)), flush=True)
                print("MISMATCH:", err, flush=True)
                return 1
            continue
        if a.focus == "genasm":
            # GenASM (parity unpinned: checked against oracle/genasm_oracle.c): any length, error rates up to 45 % (windows beyond the
            # 16-level fast path), unrelated texts, non-ACGT bytes
            l = rng.choice([1, 2, 17, 39, 40, 41, 63, 64, 65, 100, 128, 200, 500, 1000, 3000, 8000])
            e = rng.choice([0.0, 0.01, 0.05, 0.10, 0.20, 0.30, 0.45])
            rs = ((int(l * (1 + e)) + 8 + 7) // 8) * 8
            n = int(min(rng.choice([1, 3, 64, 65, 500, 2000]), max(1, 4e7 // (l * 64))))
            bt = rng.random() < 0.7
            params = engine.make_params("genasm", 0, rs, backtrace=bt, res8=(not bt and rng.random() < 0.3))
            req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
            for i in range(n):
                r = rng.random()
                if r < 0.1:                                   # unrelated text
                    tl = int(req["text_len"][i]); txt[i, :tl] = np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(tl)), dtype=np.uint8)
                elif r < 0.2:
                    pat[i, rng.randrange(l)] = ord(rng.choice("Nn*"))
            os.environ.pop("AIM_GA_PER_CU", None)
            ga = rng.choice(["", "", "1", "9", "32"])         # residency: the default, or another number of wavefronts per CU
            if ga: os.environ["AIM_GA_PER_CU"] = ga
            case = dict(algo="genasm", l=l, e=e, n=n, read_size=rs, backtrace=int(bt), kernel="genasm_wave_kernel", ga_per_cu=ga)
            try:
                if params.flags & capi.FLAG_RES8:
                    res, _ = engine.align(params, req, pat, txt)
                    ores, _, _ = oracle.align_batch(oracle.params("genasm", 0, rs), req["pattern_len"], req["text_len"], pat, txt, nthreads=64)
                    err = None if np.array_equal(res["score"], ores["score"]) else "score differs (res8)"
                else:
                    err = compare("genasm", params, req, pat, txt)
            except Exception as ex:
                err = "exception: %r" % (ex,)
            cases += 1
            kernels["genasm_wave_kernel"] = kernels.get("genasm_wave_kernel", 0) + 1
            if err:
                print(json.dumps(dict(case, ok=False)), flush=True)
                print("MISMATCH:", err, flush=True)
                os.makedirs("gpurun_out", exist_ok=True)   # the failing batch, for a replay
                np.savez_compressed("gpurun_out/fuzz_genasm_fail.npz", req=req, pat=pat, txt=txt, rs=rs, bt=int(bt), ga=ga)
                return 1
            continue
        if a.focus == "fused":
            # packed rows in, {idx, score} or compact CIGAR out, through aim_set_submit / aim_set_wait; text against the oracle's
            for k in list(os.environ):
                if k.startswith("AIM_") and k != "AIM_LIB" and not k.startswith("AIM_DEBUG_POISON"): os.environ.pop(k)
            if rng.random() < 0.5:      # lane shapes
                rs = rng.choice([80, 112, 136, 144, 160, 176]); l = rng.randint(1, rs - 12); ms = rng.randint(0, 10); cost = {}
                e = rng.choice([0.0, 0.01, 0.02, 0.05, 0.10])
            else:                       # group shapes
                l = rng.choice([20, 64, 100, 150, 250, 400, 1000]); e = rng.choice([0.02, 0.05, 0.10])
                cost = dict(mismatch=rng.randint(1, 9), gap_o=rng.randint(1, 9), gap_e=rng.randint(1, 4)) if rng.random() < 0.4 else {}
                ms, rs = engine.launcher_sizes("wfa", l, e, **cost)
                if rng.random() < 0.2: ms = max(1, ms // 2)
                if rng.random() < 0.2: os.environ["AIM_GROUP_G"] = rng.choice(["1", "4", "16", "64"])
                if rng.random() < 0.1: os.environ["AIM_GROUP_WLDS"] = "64"
            if l + int(np.ceil(l * e)) + 1 > rs: e = 0.0
            n = int(min(rng.choice([1, 64, 65, 1000, 5000]), max(1, a.max_cells // (l * l))))
            bt = rng.random() < 0.7
            params = engine.make_params("wfa", ms, rs, backtrace=bt, reduce=rng.random() < 0.7, req8=True, res8=not bt, **cost)
            req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
            for _ in range(rng.choice([0, 0, 1, 4])):
                pat[rng.randrange(n), rng.randrange(max(1, l))] = ord(rng.choice("Nn*"))
            op = oracle.params("wfa", ms, rs, backtrace=bt, reduce=bool(params.flags & capi.FLAG_REDUCE), mismatch=params.mismatch, gap_o=params.gap_o, gap_e=params.gap_e)
            ores, oops, _ = oracle.align_batch(op, req["pattern_len"], req["text_len"], pat, txt, nthreads=32)
            err = None
            try:
                with engine.DeviceSet(1) as st:
                    cap = n * (2 * min(ms, l + 8) + 12) + 64
                    st.configure_slots(params, n, slots=1, max_raw=n, max_runs=(cap if bt else 0))
                    st.submit(0, 0, req, packed=engine.pack_batch(req, pat, txt), cigar_runs_cap=(cap if bt else 0))
                    out = st.wait(0, 0, check=False)
                    kn = st.plan_describe(0).split()[0]
                if bt:
                    if not np.array_equal(out["cig"]["score"], ores["score"]): err = "score differs"
                    elif (ores["status"] == 0).all() and engine.format_output_runs(out["cig"], out["runs"]) != oracle.format_output(ores, oops, True): err = "CIGAR text differs"
                elif not np.array_equal(out["res"]["score"], ores["score"]): err = "score differs"
            except Exception as ex:
                err = "exception: %r" % (ex,); kn = "?"
            cases += 1
            kernels[kn] = kernels.get(kn, 0) + 1
            if err:
                print(json.dumps(dict(algo="wfa", l=l, e=e, n=n, max_score=ms, read_size=rs, backtrace=int(bt), cost=cost, kernel=kn, env={k: v for k, v in os.environ.items() if k.startswith("AIM_")}, ok=False)), flush=True)
                print("MISMATCH:", err, flush=True)
                return 1
            continue
        algo = "wfa" if a.focus == "wfa" else (rng.choice(["nw", "swg"]) if a.focus == "dp" else rng.choice(["wfa", "wfa", "wfa", "nw", "swg"]))
        l = rng.choice([330, 400, 700, 1000, 1500, 2500, 2800, 3500, 5000, 6000, 7500]) if a.focus == "dp" else rng.choice([3, 8, 20, 33, 64, 100, 100, 150, 250, 300, 400, 700, 1000, 1500, 2500, 3500])
        e = rng.choice([0.0, 0.01, 0.02, 0.05, 0.10, 0.15, 0.25])
        cost = {}
        if rng.random() < 0.5:
            if algo == "wfa": cost = dict(mismatch=rng.randint(1, 9), gap_o=rng.randint(1, 9), gap_e=rng.randint(1, 4))
            elif algo == "swg": cost = dict(mismatch=rng.randint(1, 9), gap_o=rng.randint(1, 9), gap_e=rng.randint(1, 4))
            else: cost = dict(mismatch=rng.randint(1, 9), gap=rng.randint(1, 9))
        try:
            ms, rs = engine.launcher_sizes(algo, l, e, **cost)
            if algo == "nw" and cost and rng.random() < 0.5:        # GAP_I != GAP_D on the long-read NW kernels
                cost = dict(mismatch=cost["mismatch"], gap_i=rng.randint(1, 9), gap_d=rng.randint(1, 9))
        except Exception:
            continue
        r0 = rng.random()
        if algo == "wfa" and r0 < 0.2: ms = max(1, ms // 2)          # some pairs exceed MAX_SCORE
        elif algo == "wfa" and r0 < 0.3: ms = max(0, rng.randint(0, 6))     # tiny caps (the static kernel's territory)
        elif r0 < 0.4: ms = min(2 * ms + 1, 600)
        n = int(min(rng.choice([1, 7, 64, 65, 300, 1000, 5000]), max(1, a.max_cells // (l * l))))
        kw = dict(backtrace=rng.random() < 0.6, **cost)
        if algo == "wfa": kw["reduce"] = rng.random() < 0.6
        if algo == "swg" and rng.random() < 0.3: kw["swg_w16"] = True
        try:
            params = engine.make_params(algo, ms, rs, **kw)
        except Exception:
            continue
        env = {}
        r = rng.random()
        if algo == "wfa" and r < 0.25: env["AIM_GROUP_G"] = rng.choice(["1", "2", "4", "8", "16", "64"])
        if algo == "wfa" and 0.25 <= r < 0.35: env["AIM_FORCE_WAVE"] = "1"
        if algo != "wfa" and r < 0.25: env["AIM_DPW_NW"] = rng.choice(["1", "2", "4"])
        if algo != "wfa" and 0.25 <= r < 0.4: env["AIM_STRIP_K"] = rng.choice(["16", "20", "24", "32"])
        if algo != "wfa" and 0.7 <= r < 0.75: env["AIM_DPW_LEGACY"] = "1"
        if algo != "wfa" and 0.4 <= r < 0.5: env["AIM_FORCE_DPWAVE"] = "1"
        if algo != "wfa" and 0.5 <= r < 0.6: env["AIM_DPL_SEQ_LDS"] = "0"
        if algo != "wfa" and rng.random() < 0.3: env["AIM_DPL_NO_REG"] = "1"      # short reads: pattern row from the LDS image / global memory instead of registers
        if algo != "wfa" and 0.6 <= r < 0.7: env["AIM_DPL_PER_CU"] = rng.choice(["1", "3", "12"])
        if algo == "wfa" and 0.35 <= r < 0.45: env["AIM_GROUP_PER_CU"] = rng.choice(["1", "5", "32"])
        if algo == "wfa" and 0.45 <= r < 0.5: env.update(AIM_FORCE_WAVE="1", AIM_WFA_NO_RING="1")
        if algo == "wfa" and rng.random() < 0.15: env["AIM_GROUP_WLDS"] = rng.choice(["64", "80", "96", "112"])   # ring rows that pairs outgrow / rows that are no power of two
        if rng.random() < 0.15: env["AIM_SCRATCH_GB"] = rng.choice(["0.25", "0.5", "2"])
        for k in ("AIM_GROUP_G", "AIM_FORCE_WAVE", "AIM_DPW_NW", "AIM_FORCE_DPWAVE", "AIM_DPL_SEQ_LDS", "AIM_DPL_PER_CU", "AIM_GROUP_PER_CU",
                  "AIM_WFA_NO_RING", "AIM_SCRATCH_GB", "AIM_STRIP_K", "AIM_DPW_LEGACY", "AIM_GROUP_WLDS", "AIM_DPL_NO_REG"): os.environ.pop(k, None)
        os.environ.update(env)
        req, pat, txt = engine.gen_pairs(rng.randint(1, 1 << 30), 0, n, l, e, rs)
        if n > 3 and rng.random() < 0.3: pat[rng.randrange(n), rng.randrange(max(1, l // 2))] = ord("N")   # non-ACGT byte
        if algo != "wfa" and l >= 1500 and rng.random() < 0.3:     # paths that shift by many diagonals half-way (a block missing from the text, other bases appended): the banded
            blk = rng.choice([l // 5, l // 3, 700, 1300])          # direction bits of dp_strip (K = 20) must notice and fill the pair again
            for i in range(0, n, rng.choice([1, 2, 5])):
                pl = int(req["pattern_len"][i])
                if pl < 2 * blk + 10: continue
                p_ = pat[i, :pl]
                cut = rng.randint(1, pl - blk - 1)
                t_ = np.concatenate([p_[:cut], p_[cut + blk:], np.frombuffer(bytes(rng.choice(b"ACGT") for _ in range(blk)), dtype=np.uint8)])[:rs]
                txt[i, :] = 0
                txt[i, :len(t_)] = t_
                req["text_len"][i] = len(t_)
            if rng.random() < 0.5: env["AIM_STRIP_K"] = "20"; os.environ["AIM_STRIP_K"] = "20"
        if algo != "wfa" and rng.random() < 0.3:                   # length outliers: tails of any size, plen > 2 tlen (the last row's tail wraps the flat table more than once), one-character texts
            for _ in range(rng.choice([1, 3, 20])):
                i = rng.randrange(n)
                pl = int(req["pattern_len"][i])
                req["text_len"][i] = max(1, rng.choice([pl - rng.randint(1, 70), pl // 2, pl // 3, pl // 7, pl // 40, 1]))
        kn = lib.aim_kernel_name(C.byref(params)).decode()
        case = dict(algo=algo, l=l, e=e, n=n, max_score=ms, read_size=rs, kernel=kn, env=env, **{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
        try:
            err = compare(algo, params, req, pat, txt, allow_nomem="AIM_SCRATCH_GB" in env)
            nomem_pairs += getattr(compare, "nomem", 0)
        except Exception as ex:
            err = "exception: %r" % (ex,)
            if "AIM_SCRATCH_GB" in env and "error -3" in err:      # the documented loud AIM_ENOMEM under a forced tiny bound
                skipped += 1
                continue
        cases += 1
        kernels[kn] = kernels.get(kn, 0) + 1
        print(json.dumps(dict(case, ok=err is None)), flush=True)
        if err:
            print("MISMATCH:", err, flush=True)
            return 1
    print(json.dumps({"cases": cases, "seconds": round(time.time() - t0, 1), "kernels": kernels, "enomem_under_forced_bound": skipped, "pair_nomem_under_forced_bound": nomem_pairs, "all_ok": True}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
