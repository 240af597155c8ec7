#!/usr/bin/env python3
"""Diagnostic: read the per-segment s_memtime sums of a -DAIM_LANE_STAMPS=1 build (never quote its run time)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import capi, engine
lib = capi.load()
n = 1 << 22
ms, rs = engine.launcher_sizes("wfa", 100, 0.01)
params = engine.make_params("wfa", ms, rs, reduce=True, req8=True, res8=True)
req, pat, txt = engine.gen_pairs(42, 0, n, 100, 0.01, rs)
dev = torch.device("cuda", 0)
def to_dev(a, pad=64):
    t = torch.zeros(a.nbytes + pad, dtype=torch.uint8, device=dev); t[:a.nbytes].copy_(torch.from_numpy(a.view(np.uint8).reshape(-1))); return t
d_req, d_pat, d_txt = to_dev(engine.to_request8(req)), to_dev(pat), to_dev(txt)
d_res = torch.zeros(n * 24 + 64, dtype=torch.uint8, device=dev)
sb = lib.aim_scratch_bytes(C.byref(params), n)
d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
for _ in range(3):
    capi.check(lib.aim_align_device(C.byref(params), n, d_req.data_ptr(), d_pat.data_ptr(), d_txt.data_ptr(), d_res.data_ptr(), None, d_scr.data_ptr(), sb, None))
torch.cuda.synchronize()
todo_bytes = 256   # round 2: the lane kernel has no to-do region; stamps sit behind the first 256 bytes
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
st = d_scr[todo_bytes: todo_bytes + grid * 64].cpu().numpy().view(np.uint64).reshape(grid, 8).astype(np.float64)
names = ["loop/store", "wait DMA", "LDS reads", "DMA issue", "pack", "diag+WFA", "store issue", "-"]
groups = n / 64 / grid
tot = st.sum(axis=1).mean()
print("per-wave total ticks %.0f, groups/wave %.1f, ticks/group %.0f" % (tot, groups, tot / groups))
for i, nm in enumerate(names):
    print("%-12s %8.0f ticks/group  %5.1f%%" % (nm, st[:, i].mean() / groups, 100 * st[:, i].mean() / tot))
