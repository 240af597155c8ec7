#!/bin/bash
# round-2 final evidence run: profiles (kernel stats + PMC) for every BASELINE kernel, one more soak, headline bench
O=gpurun_out/r02z; mkdir -p $O profiles/r02
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out profiles/r02/wfa_lane_pmc_summary.json --kernel wfa_lane_kernel --pairs 4194304 --alg-bytes 905968812 --fetch-x2 --io compact \
   --note "Cross-check: 4194304 pairs x (224 B rows + 8 B request) = 973.1 MB read, x 8 B result = 33.6 MB written." \
   -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/pmc_lane.log 2>&1; tail -1 $O/pmc_lane.log
python3 tools/pmc_summary.py --out profiles/r02/wfa_group_pmc_summary.json --kernel wfa_group_kernel --pairs 65536 \
   --note "cfg3: WFA-adaptive l=1000 e=5% with CIGAR, 65536 pairs, G=32 (two pairs per wavefront), LDS ring rows of 128; int16 LDS/HBM traffic: FETCH_SIZE kept raw (uncalibrated width)." \
   -- python3 tools/bench_configs.py wfa_l1000_e5_cigar > $O/pmc_group.log 2>&1; tail -1 $O/pmc_group.log
python3 tools/pmc_summary.py --out profiles/r02/dp_wave_pmc_summary.json --kernel dp_wave_kernel --pairs 256 \
   --note "cfg4: SWG l=10000 e=1% with CIGAR, 256 pairs, 12 wavefronts per pair; 16-B-per-lane table stores (WRITE_SIZE exact), mixed-width reads: FETCH_SIZE kept raw." \
   -- python3 tools/bench_configs.py swg_l10000_e1_cigar_n256 > $O/pmc_dpw.log 2>&1; tail -1 $O/pmc_dpw.log
python3 tools/pmc_summary.py --out profiles/r02/genasm_wave_pmc_summary.json --kernel genasm_wave_kernel --pairs 1024 \
   --note "cfg5 (parity unpinned): GenASM l=100000 e=10% with CIGAR, 1024 pairs, one pair per wavefront; byte-granular loads/stores: FETCH_SIZE kept raw." \
   -- python3 tools/bench_configs.py genasm_l100000_e10_cigar > $O/pmc_genasm.log 2>&1; tail -1 $O/pmc_genasm.log
timeout 400 python tools/soak_dp_wave.py --seconds 200 --slots 8 --seed 7 > $O/soak_more.json 2> $O/soak_more.err; echo "soak rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --backtrace --pairs 1048576 --no-cpu-baseline > $O/bench_cigar.json 2> $O/bench_cigar.err
cp -r profiles/r02 $O/
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('%.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle'], 'traffic', d['roofline']['traffic'], 'e2e packed %.3g ascii %.3g' % (d['e2e']['packed']['pairs_per_s'], d['e2e']['ascii']['pairs_per_s']))
d=json.load(open('$O/soak_more.json')); print('soak', d['launches'], d['failed_slot'])
"
