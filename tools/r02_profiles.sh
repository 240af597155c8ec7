#!/bin/bash
# round-2 profile run: rocprofv3 kernel stats + PMC summaries for every BASELINE configuration's kernel, kernel timers of all configs
O=gpurun_out/r02p; mkdir -p $O profiles/r02
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out profiles/r02/wfa_lane_pmc_summary.json --kernel wfa_lane_kernel --pairs 4194304 --alg-bytes 905968812 --fetch-x2 --io compact \
   --note "Cross-check: 4194304 pairs x (224 B rows + 8 B request) = 973.1 MB read, x 8 B result = 33.6 MB written." \
   -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_lane.log 2>&1; tail -1 $O/pmc_lane.log
python3 tools/pmc_summary.py --out profiles/r02/wfa_group_pmc_summary.json --kernel wfa_group_kernel --pairs 65536 --skip-first 0 \
   --note "cfg3: WFA-adaptive l=1000 e=5% with CIGAR, 65536 pairs, G=64 (one pair per wavefront); per-lane int16 LDS/HBM traffic: FETCH_SIZE kept raw (uncalibrated width)." \
   -- python3 tools/bench_configs.py wfa_l1000_e5_cigar > $O/pmc_group.log 2>&1; tail -1 $O/pmc_group.log
python3 tools/pmc_summary.py --out profiles/r02/dp_wave_pmc_summary.json --kernel dp_wave_kernel --pairs 256 \
   --note "cfg4: SWG l=10000 e=1% with CIGAR, 256 pairs, 12 wavefronts per pair; 16-B-per-lane table stores (WRITE_SIZE exact), mixed-width reads: FETCH_SIZE kept raw." \
   -- python3 tools/bench_configs.py swg_l10000_e1_cigar_n256 > $O/pmc_dpw.log 2>&1; tail -1 $O/pmc_dpw.log
python3 tools/bench_configs.py > profiles/r02/all_configs_kernel_timers.jsonl 2> $O/configs.err; cat profiles/r02/all_configs_kernel_timers.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('%-28s %-18s %10.4g pairs/s %8.1f GCUPS %8.1f GB/s' % (d['config'], d['kernel'], d['pairs_per_s'], d['gcups'], d['algorithmic_GBps']))
"
cp -r profiles/r02 $O/
