#!/bin/bash
O=gpurun_out/r02g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python tools/bench_configs.py > $O/all_configs_kernel_timers.jsonl 2> $O/configs.err; python3 -c "
import sys, json
for l in open('$O/all_configs_kernel_timers.jsonl'):
    d=json.loads(l); print('%-30s %-18s %10.4g pairs/s %9.1f GCUPS %8.1f GB/s' % (d['config'], d['kernel'], d['pairs_per_s'], d['gcups'], d['algorithmic_GBps']))
"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out profiles/r02/wfa_group_pmc_summary.json --kernel wfa_group_kernel --pairs 65536 \
   --note "cfg3: WFA-adaptive l=1000 e=5% with CIGAR, 65536 pairs, G=32 (two pairs per wavefront), LDS ring rows of 128, 20 workgroups per CU; int16 LDS/HBM traffic: FETCH_SIZE kept raw (uncalibrated width)." \
   -- python3 tools/bench_configs.py wfa_l1000_e5_cigar > $O/pmc_group.log 2>&1; tail -1 $O/pmc_group.log
python3 tools/pmc_summary.py --out profiles/r02/genasm_wave_pmc_summary.json --kernel genasm_wave_kernel --pairs 1024 \
   --note "cfg5 (parity unpinned): GenASM l=100000 e=10% with CIGAR, 1024 pairs, one pair per wavefront; byte-granular loads/stores: FETCH_SIZE kept raw." \
   -- python3 tools/bench_configs.py genasm_l100000_e10_cigar > $O/pmc_genasm.log 2>&1; tail -1 $O/pmc_genasm.log
cp profiles/r02/wfa_group* profiles/r02/genasm_wave* $O/
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('%.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle'], 'traffic', d['roofline']['traffic'], 'e2e packed %.3g ascii %.3g' % (d['e2e']['packed']['pairs_per_s'], d['e2e']['ascii']['pairs_per_s']))
"
