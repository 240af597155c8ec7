#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of a -DAIM_GROUP_STAMPS=1 build on BASELINE config 3 (score-only)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine
lib = capi.load()
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
E = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 15
ms, rs = engine.launcher_sizes("wfa", L, E)
params = engine.make_params("wfa", ms, rs, reduce=True)
assert lib.aim_kernel_name(C.byref(params)) == b"wfa_group_kernel"
req, pat, txt = engine.gen_pairs(42, 0, n, L, E, rs)
dev = torch.device("cuda", 0)
def to_dev(a, pad=64):
    t = torch.zeros(a.nbytes + pad, dtype=torch.uint8, device=dev); t[:a.nbytes].copy_(torch.from_numpy(a.view(np.uint8).reshape(-1))); return t
d_req, d_pat, d_txt = to_dev(req), to_dev(pat), to_dev(txt)
d_res = torch.zeros(n * 24 + 64, dtype=torch.uint8, device=dev)
sb = lib.aim_scratch_bytes(C.byref(params), n)
d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
capi.check(lib.aim_align_device(C.byref(params), n, d_req.data_ptr(), d_pat.data_ptr(), d_txt.data_ptr(), d_res.data_ptr(), None, d_scr.data_ptr(), sb, None))
torch.cuda.synchronize()
buf = C.create_string_buffer(512); capi.check(lib.aim_plan_describe(C.byref(params), n, buf, 512)); plan = buf.value.decode(); print(plan)
grid = int(plan.split("grid=")[1].split()[0]) if len(sys.argv) < 2 or sys.argv[1] == "auto" else int(sys.argv[1])
ppw = 64 // int(plan.split(" G=")[1].split()[0])
st = d_scr[256: 256 + grid * 64].cpu().numpy().view(np.uint64).reshape(grid, 8).astype(np.float64)
res = np.frombuffer(d_res[: n * 24].cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
steps = float(res["score"].sum()) / grid / ppw   # wave score-steps ~ mean score x units per wave (the wave steps to its slowest pair: a lower bound)
names = ["staging+pack", "reduce+desc+end", "score++/src desc", "compute+extend", "desc store", "loop edge", "exit", "backtrace+result"]
tot = st.sum(axis=1).mean()
print("per-wave ticks %.0f, score steps per wave %.0f, ticks/step %.0f" % (tot, steps, tot / steps))
for i, nm in enumerate(names):
    print("%-13s %9.0f ticks/step %5.1f%%" % (nm, st[:, i].mean() / steps, 100 * st[:, i].mean() / tot))
