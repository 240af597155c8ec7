"""Re-run a GenASM batch that tools/fuzz_parity.py --focus genasm dumped (gpurun_out/fuzz_genasm_fail.npz) in a fresh process and
write the first differing pair -- sequences, both op strings -- to gpurun_out/ga_debug.json (diagnostic; run on the GPU box)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aim_amd import engine
from oracle import oracle

z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_genasm_fail.npz")
req, pat, txt, rs = z["req"], z["pat"], z["txt"], int(z["rs"])
if str(z["ga"]): os.environ["AIM_GA_PER_CU"] = str(z["ga"])
n = len(req)
params = engine.make_params("genasm", 0, rs, backtrace=True)
res, ops = engine.align(params, req, pat, txt, check=False)
ores, oops, _ = oracle.align_batch(oracle.params("genasm", 0, rs, backtrace=True), req["pattern_len"], req["text_len"], pat, txt, nthreads=64)
bad = [i for i in range(n) if res["score"][i] != ores["score"][i] or not np.array_equal(ops[i, :res["end_offset"][i]], oops[i, :ores["end_offset"][i]])]
print("pairs", n, "differing", len(bad), bad[:10], flush=True)
if bad:
    i = bad[0]
    pl, tl = int(req["pattern_len"][i]), int(req["text_len"][i])
    out = dict(pair=i, p=bytes(pat[i, :pl]).decode("latin1"), t=bytes(txt[i, :tl]).decode("latin1"),
               hip=bytes(ops[i, :res["end_offset"][i]]).decode("latin1"), ora=bytes(oops[i, :ores["end_offset"][i]]).decode("latin1"),
               hip_score=int(res["score"][i]), ora_score=int(ores["score"][i]), nbad=len(bad))
    json.dump(out, open("gpurun_out/ga_debug.json", "w"))
