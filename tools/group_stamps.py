#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of a -DAIM_GROUP_STAMPS=1 build on BASELINE config 3 (score-only)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine
lib = capi.load()
n = 1 << 15
ms, rs = engine.launcher_sizes("wfa", 1000, 0.05)
params = engine.make_params("wfa", ms, rs, reduce=True)
assert lib.aim_kernel_name(C.byref(params)) == b"wfa_group_kernel"
req, pat, txt = engine.gen_pairs(42, 0, n, 1000, 0.05, rs)
dev = torch.device("cuda", 0)
def to_dev(a, pad=64):
    t = torch.zeros(a.nbytes + pad, dtype=torch.uint8, device=dev); t[:a.nbytes].copy_(torch.from_numpy(a.view(np.uint8).reshape(-1))); return t
d_req, d_pat, d_txt = to_dev(req), to_dev(pat), to_dev(txt)
d_res = torch.zeros(n * 24 + 64, dtype=torch.uint8, device=dev)
sb = lib.aim_scratch_bytes(C.byref(params), n)
d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
capi.check(lib.aim_align_device(C.byref(params), n, d_req.data_ptr(), d_pat.data_ptr(), d_txt.data_ptr(), d_res.data_ptr(), None, d_scr.data_ptr(), sb, None))
torch.cuda.synchronize()
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
st = d_scr[256: 256 + grid * 64].cpu().numpy().view(np.uint64).reshape(grid, 8).astype(np.float64)
res = np.frombuffer(d_res[: n * 24].cpu().numpy().tobytes(), dtype=capi.RESULT_DTYPE)
steps = float(res["score"].sum()) / grid
names = ["staging+pack", "reduce+desc+end", "score++/src desc", "compute+extend", "desc store", "loop edge", "exit", "backtrace+result"]
tot = st.sum(axis=1).mean()
print("per-wave ticks %.0f, score steps per wave %.0f, ticks/step %.0f" % (tot, steps, tot / steps))
for i, nm in enumerate(names):
    print("%-13s %9.0f ticks/step %5.1f%%" % (nm, st[:, i].mean() / steps, 100 * st[:, i].mean() / tot))
