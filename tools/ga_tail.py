"""GenASM long reads: how many pairs of a synthetic batch lose the diagonal (every later window takes the 64-level path)? Scores per pair
from one launch; a pair that drifted has a score of ~half its length instead of ~8.5 % (diagnostic; run on the GPU box)."""
import os, sys, time
import numpy as np
try:
    import torch   # (part 2 needs torch's streams; torch has to initialise the device before the library does)
    torch.zeros(1, device="cuda:0")
except Exception:
    torch = None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
rs = ((int(L * (1 + E)) + 8 + 7) // 8) * 8
params = engine.make_params("genasm", 0, rs)
req, pat, txt = engine.gen_pairs(42, 0, n, L, E, rs)
t0 = time.time()
res, _ = engine.align(params, req, pat, txt)
sc = res["score"].astype(np.int64)
med = np.median(sc)
out = np.nonzero(sc > 2 * med)[0]
print("pairs %d  l=%d e=%g: median score %d, max %d, pairs above twice the median: %d %s  (wall %.1f s)" % (n, L, E, med, sc.max(), len(out), [(int(i), int(sc[i])) for i in out[:12]], time.time() - t0))

# ---- part 2: a batch with a drifted pair keeps ONE wavefront busy ~6x longer than the rest. Does the next batch, launched on another stream, run under
# that tail? Device-resident batches of 4 096 pairs (score-only), batch 1 holds pair 4672; times from events on each stream and the wall over all.
if torch is not None and L == 100000 and n >= 16384:
    import ctypes as C
    from aim_amd import capi
    lib = capi.load()
    dev = torch.device("cuda:0")
    B = 4096
    nb = 4
    def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
    batches = []
    for b in range(nb):
        sl = slice(b * B, (b + 1) * B)
        r = req[sl].copy()
        d = dict(req=to_dev(r), pat=torch.cat([to_dev(pat[sl]), torch.zeros(64, dtype=torch.uint8, device=dev)]), txt=torch.cat([to_dev(txt[sl]), torch.zeros(64, dtype=torch.uint8, device=dev)]),
                 res=torch.zeros(B * capi.RESULT_DTYPE.itemsize + 64, dtype=torch.uint8, device=dev))
        need = lib.aim_scratch_bytes(C.byref(params), B)
        d["scr"] = torch.empty(int(need) + 256, dtype=torch.uint8, device=dev)
        batches.append(d)
    def launch(d, stream):
        capi.check(lib.aim_align_device(C.byref(params), B, d["req"].data_ptr(), d["pat"].data_ptr(), d["txt"].data_ptr(), d["res"].data_ptr(), None,
                                        d["scr"].data_ptr(), d["scr"].numel(), stream.cuda_stream))
    s0 = torch.cuda.current_stream(dev)
    for d in batches: launch(d, s0)
    torch.cuda.synchronize(dev)
    one = []
    for d in batches:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0); launch(d, s0); e1.record(s0); torch.cuda.synchronize(dev)
        one.append(e0.elapsed_time(e1))
    print("one stream, batch by batch (ms):", ["%.1f" % t for t in one], "sum %.1f" % sum(one))
    for ns in (2, 4):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        torch.cuda.synchronize(dev)
        t0 = time.time()
        for i, d in enumerate(batches): launch(d, streams[i % ns])
        torch.cuda.synchronize(dev)
        print("%d streams, all %d batches in flight: wall %.1f ms" % (ns, nb, (time.time() - t0) * 1e3))
