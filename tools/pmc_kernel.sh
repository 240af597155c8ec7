#!/bin/bash
# Per-kernel PMC sums for one tools/bench_configs.py config (run on the GPU box; counters in separate passes).
# usage: tools/pmc_kernel.sh <config> <kernel-substring>
cfg=$1; kern=$2
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
if [ -n "$AIM_PMC_SETS" ]; then IFS=';' read -ra SETS <<< "$AIM_PMC_SETS"; else SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES"); fi
for set in "${SETS[@]}"; do
  d=gpurun_out/pmc_$$; rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $d -o p --output-format csv -- python3 tools/bench_configs.py $cfg > /dev/null 2>&1
  python3 - "$d" "$kern" <<'PY'
import csv, glob, sys, collections
d, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-22s %.4g per launch (%d launches)" % (k, acc[k] / max(1, n[k]), n[k]))
PY
  rm -rf $d
done
