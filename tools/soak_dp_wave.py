#!/usr/bin/env python3
"""Soak test for the long-read DP kernels: dp_wave_kernel (row scan; the family of the one unexplained CLI difference of round 1:
NW, l=700, e=10 %, CIGAR, 40 pairs, 2 wavefronts per pair; DESIGN.md "open items") and, since round 3, dp_strip_kernel (the
column-strip pipeline: its wavefronts synchronise through LDS mailboxes with sequence numbers instead of barriers, which is
exactly the kind of code a launch-to-launch soak is for; slots with nw = 0 take the default plan = the strip kernel, some with a
forced AIM_STRIP_K so that 2 .. 4 wavefronts per pair exchange messages).

Every slot owns one configuration (algorithm, READ_SIZE, pair count, forced wavefronts per pair), its own HBM buffers and
its own stream. A slot's first launch is checked against the CPU oracle (checker only); every later launch of the SAME
batch is compared with that first result ON THE DEVICE (results and the whole ops buffer, which is zeroed once so that
bytes a launch does not write stay equal), so no D2H happens per launch and 10^5..10^6 launches fit a GPU budget. All
slots run concurrently (different streams), and one more stream keeps wfa_lane_kernel launches in flight, so timing,
residency and memory traffic around each launch vary. Grids are below residency (9..100 pairs = workgroups) and, for the
"big" slots, above it. On a mismatch the slot's inputs and both outputs are kept under gpurun_out/soak_fail/.

    python tools/soak_dp_wave.py [--seconds 300] [--poison-lds BYTE] [--slots 8] [--seed 1]
"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--slots", type=int, default=8)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--poison-lds", type=int, default=None, help="AIM_DEBUG_POISON_LDS byte for every launch")
ap.add_argument("--check-every", type=int, default=500)
a = ap.parse_args()
if a.poison_lds is not None:
    os.environ["AIM_DEBUG_POISON_LDS"] = str(a.poison_lds)

import torch
from aim_amd import capi, engine
from oracle import oracle
lib = capi.load()
dev = torch.device("cuda", 0)
rng = np.random.RandomState(a.seed)


def to_dev(arr, pad=64):
    t = torch.zeros(arr.nbytes + pad, dtype=torch.uint8, device=dev)
    t[: arr.nbytes].copy_(torch.from_numpy(arr.view(np.uint8).reshape(-1)))
    return t


class Slot:
    def __init__(self, algo, l, e, n, nw, seed, extra_env=None):
        self.algo, self.l, self.e, self.n, self.nw, self.seed = algo, l, e, n, nw, seed
        ms, rs = engine.launcher_sizes(algo, l, e)
        self.rs = rs
        self.params = engine.make_params(algo, ms, rs, backtrace=True)
        self.req, self.pat, self.txt = engine.gen_pairs(seed, 0, n, l, e, rs)
        self.d_req, self.d_pat, self.d_txt = to_dev(self.req), to_dev(self.pat), to_dev(self.txt)
        self.d_res = torch.zeros(n * capi.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.d_ops = torch.zeros(n * 2 * rs + 64, dtype=torch.uint8, device=dev)
        self.env = {"AIM_DPW_NW": str(nw)} if nw else dict(extra_env or {})
        self._setenv()
        self.scratch = torch.zeros(max(256, lib.aim_scratch_bytes(C.byref(self.params), n)), dtype=torch.uint8, device=dev)
        buf = C.create_string_buffer(512)
        capi.check(lib.aim_plan_describe(C.byref(self.params), n, buf, 512))
        self.plan = buf.value.decode()
        self.stream = torch.cuda.Stream(device=dev)
        self.bad = torch.zeros(1, dtype=torch.int64, device=dev)
        self.launches = 0
        self.ref_res = self.ref_ops = None

    def _setenv(self):
        os.environ.pop("AIM_DPW_NW", None)
        os.environ.pop("AIM_STRIP_K", None)
        os.environ.update(self.env)

    def launch(self):
        self._setenv()
        capi.check(lib.aim_align_device(C.byref(self.params), self.n, self.d_req.data_ptr(), self.d_pat.data_ptr(),
                                        self.d_txt.data_ptr(), self.d_res.data_ptr(), self.d_ops.data_ptr(),
                                        self.scratch.data_ptr(), self.scratch.numel(), self.stream.cuda_stream))
        self.launches += 1

    def first(self):
        self.launch()
        self.stream.synchronize()
        res = np.frombuffer(self.d_res.cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
        ops = self.d_ops[: self.n * 2 * self.rs].cpu().numpy().reshape(self.n, 2 * self.rs)
        op = oracle.params(self.algo, self.params.max_score, self.rs, backtrace=True)
        ores, oops, _ = oracle.align_batch(op, self.req["pattern_len"], self.req["text_len"], self.pat, self.txt, nthreads=1)
        for f in ("score", "status", "begin_offset", "end_offset"):
            assert np.array_equal(res[f], ores[f]), (self.plan, f)
        for i in range(self.n):
            if res["status"][i] == 0:
                b, e = int(res["begin_offset"][i]), int(res["end_offset"][i])
                assert np.array_equal(ops[i, b:e], oops[i, b:e]), (self.plan, i)
        self.ref_res, self.ref_ops = self.d_res.clone(), self.d_ops.clone()

    def again(self):
        self.launch()
        with torch.cuda.stream(self.stream):
            self.bad += (self.d_res != self.ref_res).any().to(torch.int64) + (self.d_ops != self.ref_ops).any().to(torch.int64)

    def dump(self, tag):
        keep = os.path.join(ROOT, "gpurun_out", "soak_fail", tag)
        os.makedirs(keep, exist_ok=True)
        np.save(os.path.join(keep, "req.npy"), self.req); np.save(os.path.join(keep, "pat.npy"), self.pat); np.save(os.path.join(keep, "txt.npy"), self.txt)
        for name, t in (("res", self.d_res), ("ops", self.d_ops), ("ref_res", self.ref_res), ("ref_ops", self.ref_ops)):
            np.save(os.path.join(keep, name + ".npy"), t.cpu().numpy())
        json.dump(dict(plan=self.plan, algo=self.algo, l=self.l, e=self.e, n=self.n, nw=self.nw, seed=self.seed, launches=self.launches),
                  open(os.path.join(keep, "case.json"), "w"))


# the failing family first (NW l=700 e=10 % 40 pairs, 2 wavefronts per pair), then its neighbours
menu = [("nw", 700, 0.10, 40, 2, None), ("swg", 3000, 0.02, 24, 0, {"AIM_STRIP_K": "16"}), ("nw", 700, 0.10, 9, 4, None), ("swg", 700, 0.10, 40, 0, None),
        ("nw", 3500, 0.02, 12, 0, {"AIM_STRIP_K": "16"}), ("swg", 1000, 0.05, 40, 4, None), ("nw", 700, 0.10, 100, 2, None), ("swg", 2500, 0.03, 40, 0, None),
        ("nw", 700, 0.10, 3000, 2, None), ("swg", 700, 0.05, 9, 1, None), ("nw", 1000, 0.05, 40, 0, None), ("swg", 1000, 0.05, 9, 2, None),
        ("nw", 700, 0.10, 40, 1, None), ("swg", 700, 0.10, 40, 2, None), ("nw", 1000, 0.05, 100, 2, None), ("nw", 300, 0.10, 40, 2, None)]
slots = []
for i in range(a.slots):
    algo, l, e, n, nw, extra = menu[i % len(menu)]
    slots.append(Slot(algo, l, e, n, nw, int(rng.randint(1, 1 << 30)), extra))
for s in slots:
    s.first()

# background traffic: the headline kernel on its own stream
ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
bg_params = engine.make_params("wfa", ms, rs, reduce=True)
bg_n = 1 << 20
breq, bpat, btxt = engine.gen_pairs(99, 0, bg_n, 100, 0.01, rs)
b_req, b_pat, b_txt = to_dev(breq), to_dev(bpat), to_dev(btxt)
b_res = torch.zeros(bg_n * capi.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
os.environ.pop("AIM_DPW_NW", None)
os.environ.pop("AIM_STRIP_K", None)
b_scratch = torch.zeros(max(256, lib.aim_scratch_bytes(C.byref(bg_params), bg_n)), dtype=torch.uint8, device=dev)
bg_stream = torch.cuda.Stream(device=dev)

t0 = time.time()
rounds = 0
failed = None
while time.time() - t0 < a.seconds and failed is None:
    for _ in range(a.check_every):
        for s in slots:
            s.again()
        if rounds % 4 == 0:
            capi.check(lib.aim_align_device(C.byref(bg_params), bg_n, b_req.data_ptr(), b_pat.data_ptr(), b_txt.data_ptr(), b_res.data_ptr(),
                                            None, b_scratch.data_ptr(), b_scratch.numel(), bg_stream.cuda_stream))
        rounds += 1
    torch.cuda.synchronize(dev)
    for i, s in enumerate(slots):
        if int(s.bad.item()) != 0:
            failed = i
            s.dump("slot%d" % i)
            break
total = sum(s.launches for s in slots)
print(json.dumps(dict(tool="soak_dp_wave", seconds=round(time.time() - t0, 1), launches=total, rounds=rounds, poison_lds=a.poison_lds,
                      failed_slot=failed, slots=[dict(plan=s.plan, launches=s.launches, bad=int(s.bad.item())) for s in slots])), flush=True)
sys.exit(1 if failed is not None else 0)
