#!/bin/bash
# tools/long_fuzz.sh <seconds per focus> <out.json>: one long session of every fuzzer + the launch-to-launch soak on the GPU box;
# the summary lines are collected into one JSON file (profiles/rNN/fuzz_runs.json).
S=${1:-300}; OUT=${2:-gpurun_out/fuzz_runs.json}; mkdir -p $(dirname $OUT)
echo "[" > $OUT
for f in all lane wfa fused dp dplane dpgroup genasm; do
  timeout $((S + 120)) python tools/fuzz_parity.py --seconds $S --focus $f --seed $((RANDOM)) 2>&1 | tail -1 | sed "s/^{/{\"focus\": \"$f\", /" >> $OUT; echo "," >> $OUT
done
# the same fuzzers with LDS poisoned at kernel entry: a result that depends on what LDS held before is a missing ordering (round 4: GenASM)
for f in genasm dplane dpgroup wfa; do
  AIM_DEBUG_POISON_LDS=165 timeout $((S + 120)) python tools/fuzz_parity.py --seconds $((S / 2)) --focus $f --seed $((RANDOM)) 2>&1 | tail -1 | sed "s/^{/{\"focus\": \"$f\", \"lds_poison\": 165, /" >> $OUT; echo "," >> $OUT
done
timeout $((S + 300)) python tools/fuzz_cli.py --seconds $S 2>&1 | tail -1 >> $OUT; echo "," >> $OUT
timeout $((S + 300)) python tools/soak_dp_wave.py --seconds $S --slots 8 | tail -1 >> $OUT
echo "]" >> $OUT
tail -c 1500 $OUT
