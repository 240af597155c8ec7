#!/bin/bash
# round-2 evidence refresh after the wfa_group (32-base extend) and genasm_wave (skewed DC, one-round-trip TB) changes
O=gpurun_out/r02y; mkdir -p $O profiles/r02
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out profiles/r02/wfa_group_pmc_summary.json --kernel wfa_group_kernel --pairs 65536 \
   --note "cfg3: WFA-adaptive l=1000 e=5% with CIGAR, 65536 pairs, G=32 (two pairs per wavefront), LDS ring rows of 128; int16 LDS/HBM traffic: FETCH_SIZE kept raw (uncalibrated width)." \
   -- python3 tools/bench_configs.py wfa_l1000_e5_cigar > $O/pmc_group.log 2>&1; tail -1 $O/pmc_group.log
python3 tools/pmc_summary.py --out profiles/r02/wfa_group_l100_e5_pmc_summary.json --kernel wfa_group_kernel --pairs 1048576 \
   --note "WFA-adaptive l=100 e=5% score-only, 1048576 pairs, G=8 (eight pairs per wavefront), 16 workgroups per CU: the VALU-issue-bound row (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs vs the kernel time)." \
   -- python3 tools/bench_configs.py wfa_l100_e5_score > $O/pmc_group_e5.log 2>&1; tail -1 $O/pmc_group_e5.log
python3 tools/pmc_summary.py --out profiles/r02/genasm_wave_pmc_summary.json --kernel genasm_wave_kernel --pairs 1024 \
   --note "cfg5 (parity unpinned): GenASM l=100000 e=10% with CIGAR, 1024 pairs, one pair per wavefront; byte-granular loads/stores: FETCH_SIZE kept raw." \
   -- python3 tools/bench_configs.py genasm_l100000_e10_cigar > $O/pmc_genasm.log 2>&1; tail -1 $O/pmc_genasm.log
python3 tools/bench_configs.py > profiles/r02/all_configs_kernel_timers.jsonl 2> $O/configs.err; cat profiles/r02/all_configs_kernel_timers.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('%-32s %-18s %10.4g pairs/s %8.1f GCUPS %8.1f GB/s' % (d['config'], d['kernel'], d['pairs_per_s'], d['gcups'], d['algorithmic_GBps']))
"
AIM_LIB=$PWD/build_ab/lib_gastamps.so python tools/ga_stamps.py 100000 0.10 1024 > $O/ga_stamps.txt 2>&1; cat $O/ga_stamps.txt | tail -7
cp -r profiles/r02 $O/
