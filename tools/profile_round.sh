#!/bin/bash
# tools/profile_round.sh <round-dir, e.g. r03> : the profile evidence of one round, on the GPU box (inside a gpurun call):
# rocprofv3 kernel stats + PMC summaries for every BASELINE configuration's kernel (tools/pmc_summary.py: stats pass and PMC
# passes are separate runs), then the kernel timers of every configuration of tools/bench_configs.py.
# Results go to profiles/<round-dir>/ AND are mirrored under gpurun_out/ (the only directory gpurun copies back).
R=${1:-r06}; P=profiles/$R; O=gpurun_out/profile_$R; mkdir -p $O $P
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out $P/wfa_lane_pmc_summary.json --kernel wfa_lane_kernel --pairs 4194304 --alg-bytes 905968812 --fetch-x2 --io compact \
   --note "Cross-check: 4194304 pairs x (224 B rows + 8 B request) = 973.1 MB read, x 8 B result = 33.6 MB written." \
   -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-default-io > $O/pmc_lane.log 2>&1; tail -1 $O/pmc_lane.log
python3 tools/pmc_summary.py --out $P/wfa_lane_cigar_pmc_summary.json --kernel wfa_lane_kernel --pairs 1048576 --alg-bytes 331699290 --fetch-x2 --io compact+cigar \
   --note "bench.py --backtrace --pairs 1048576: the headline kernel's CIGAR instantiation (8-byte requests in, 24-byte results + ops rows out, only the printable piece of a row prefilled); algorithmic bytes as bench.py counts them: sequence bytes + 16 B per pair + the operations produced." \
   -- python3 bench.py --backtrace --pairs 1048576 --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-default-io > $O/pmc_lane_cigar.log 2>&1; tail -1 $O/pmc_lane_cigar.log
python3 tools/pmc_summary.py --out $P/wfa_group_pmc_summary.json --kernel wfa_group_kernel --pairs 65536 \
   --note "cfg3: WFA-adaptive l=1000 e=5% with CIGAR, 65536 pairs; per-lane int16 LDS/HBM traffic: FETCH_SIZE kept raw (uncalibrated width)." \
   -- python3 tools/bench_configs.py wfa_l1000_e5_cigar > $O/pmc_group.log 2>&1; tail -1 $O/pmc_group.log
python3 tools/pmc_summary.py --out $P/dp_strip_pmc_summary.json --kernel dp_strip_kernel --pairs 256 \
   --note "cfg4: SWG l=10000 e=1% with CIGAR, 256 pairs, column-strip pipeline; round 5: four direction bits per cell (one 16-byte store per lane and row, WRITE_SIZE exact) instead of the int16 M plane." \
   -- python3 tools/bench_configs.py swg_l10000_e1_cigar_n256 > $O/pmc_dps.log 2>&1; tail -1 $O/pmc_dps.log
python3 tools/pmc_summary.py --out $P/wfa_lane_packed_pmc_summary.json --kernel wfa_lane_packed_kernel --pairs 4194304 --fetch-x2 --alg-bytes 335544320 \
   --plan "wfa_lane_packed_kernel n=4194304 (the e2e leg's plan: aim_set_plan_describe of the packed batch; bench.py prints the headline ASCII kernel's plan)" \
   --note "the drop-in path's kernel on packed batches (bench.py e2e leg, score-only): 4194304 pairs x (2 x 28 B packed rows + 8 B request) read, x 8 B written = 80 B of the PACKED wire format per pair: algorithmic bytes here are those (VERDICT r04 weak 12: the ASCII figure over a kernel that reads 2-bit rows gave 1.68 of the roof)." \
   -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_lanepk.log 2>&1; tail -1 $O/pmc_lanepk.log
python3 tools/pmc_summary.py --out $P/wfa_group_tb_pmc_summary.json --kernel wfa_group_tb_kernel --pairs 65536 \
   --note "cfg3's traceback kernel (one pair per lane over the compact per-pair history regions)." \
   -- python3 tools/bench_configs.py wfa_l1000_e5_cigar > $O/pmc_grouptb.log 2>&1; tail -1 $O/pmc_grouptb.log
python3 tools/pmc_summary.py --out $P/genasm_wave_pmc_summary.json --kernel genasm_wave_kernel --pairs 4096 \
   --note "cfg5: GenASM l=100000 e=10% with CIGAR, 4096 pairs = 16 wavefronts per CU: banded words, one DPP scan per error level, column-bound walk (parity unpinned)." \
   -- python3 tools/bench_configs.py genasm_l100000_e10_cigar_n4096 > $O/pmc_genasm.log 2>&1; tail -1 $O/pmc_genasm.log
python3 tools/pmc_summary.py --out $P/nw_reg_pmc_summary.json --kernel nw_reg_kernel --pairs 1048576 \
   --note "NW l=100 e=1% score-only, 1 Mi pairs: the DP row in registers (dp_reg.hpp); per-lane 112-byte rows read as dwords: FETCH_SIZE kept raw." \
   -- python3 tools/bench_configs.py nw_l100_e1_score > $O/pmc_nwreg.log 2>&1; tail -1 $O/pmc_nwreg.log
python3 tools/pmc_summary.py --out $P/nw_reg_cigar_pmc_summary.json --kernel nw_reg_kernel --pairs 1048576 \
   --note "NW l=100 e=1% with CIGAR, 1 Mi pairs: two direction bits per cell, round 6: only the band of four dwords around the diagonal is stored (one 16-byte unit per row and lane; 8 dwords before) and the walk fetches eight rows at a time." \
   -- python3 tools/bench_configs.py nw_l100_e1_cigar > $O/pmc_nwregc.log 2>&1; tail -1 $O/pmc_nwregc.log
python3 tools/pmc_summary.py --out $P/wfa_group_long_pmc_summary.json --kernel wfa_group_kernel --pairs 8192 \
   --note "WFA-adaptive l=10000 e=1% (MAX_SCORE 500, READ_SIZE 10112) score-only, 8192 pairs: wfa_group_kernel G=32 since round 4 (wfa_wave_kernel before)." \
   -- python3 tools/bench_configs.py wfa_l10000_e1_score > $O/pmc_grouplong.log 2>&1; tail -1 $O/pmc_grouplong.log
python3 tools/pmc_summary.py --out $P/swg_reg_pmc_summary.json --kernel swg_reg_kernel --pairs 1048576 \
   --note "SWG l=100 e=1% score-only, 1 Mi pairs: M and I rows in registers (dp_reg.hpp, round 5; int8 cells as value * 256 in 16-bit fields)." \
   -- python3 tools/bench_configs.py swg_l100_e1_score > $O/pmc_swgreg.log 2>&1; tail -1 $O/pmc_swgreg.log
python3 tools/pmc_summary.py --out $P/swg_reg_cigar_pmc_summary.json --kernel swg_reg_kernel --pairs 1048576 \
   --note "SWG l=100 e=1% with CIGAR, 1 Mi pairs: four direction bits per cell, round 6: made and stored for the band of four dwords (32 columns) around the diagonal only (14 dwords per row and lane before), the walk fetches eight rows at a time." \
   -- python3 tools/bench_configs.py swg_l100_e1_cigar > $O/pmc_swgregc.log 2>&1; tail -1 $O/pmc_swgregc.log
AIM_NO_SWG_REG=1 python3 tools/pmc_summary.py --out $P/swg_lane_pmc_summary.json --kernel swg_lane_kernel --pairs 1048576 \
   --note "swg_lane_kernel alone (AIM_NO_SWG_REG=1: what round 4 ran; now the to-do pass of swg_reg): SWG l=100 e=1% score-only, 1 Mi pairs." \
   -- python3 tools/bench_configs.py swg_l100_e1_score > $O/pmc_swglane.log 2>&1; tail -1 $O/pmc_swglane.log
AIM_NO_NW_REG=1 python3 tools/pmc_summary.py --out $P/nw_lane_pmc_summary.json --kernel nw_lane_kernel --pairs 1048576 \
   --note "nw_lane_kernel alone (AIM_NO_NW_REG=1): NW l=100 e=1% score-only, 1 Mi pairs." \
   -- python3 tools/bench_configs.py nw_l100_e1_score > $O/pmc_nwlane.log 2>&1; tail -1 $O/pmc_nwlane.log
python3 tools/pmc_summary.py --out $P/nw_reg_e5_pmc_summary.json --kernel nw_reg_kernel --pairs 1048576 \
   --note "NW l=100 e=5% score-only, 1 Mi pairs: the last row's tail cells in the kernel (round 5: no to-do pass)." \
   -- python3 tools/bench_configs.py nw_l100_e5_score > $O/pmc_nwreg5.log 2>&1; tail -1 $O/pmc_nwreg5.log
python3 tools/pmc_summary.py --out $P/dp_group_pmc_summary.json --kernel dp_group_kernel --pairs 399360 \
   --note "NW l=250 e=2% score-only, 399 360 pairs: dp_group_kernel (round 6: 20 registers = 40 cells per lane, 7 lanes per pair, 9 pairs per wavefront: the shape with the most pairs x resident wavefronts per register)." \
   -- python3 tools/bench_configs.py nw_l250_e2_score > $O/pmc_dpg.log 2>&1; tail -1 $O/pmc_dpg.log
python3 tools/pmc_summary.py --out $P/dp_group_swg_cigar_pmc_summary.json --kernel dp_group_kernel --pairs 99328 \
   --note "SWG (int16 cells) l=250 e=2% with CIGAR, 99 328 pairs: dp_group_kernel (16 registers per lane: 9 lanes per pair, 7 pairs per wavefront), four direction bits per cell + the wavefront's walks (the walker prefetches its next window)." \
   -- python3 tools/bench_configs.py swg_l250_e2_w16_cigar > $O/pmc_dpgc.log 2>&1; tail -1 $O/pmc_dpgc.log
python3 tools/pmc_summary.py --out $P/dp_group_kp20_pmc_summary.json --kernel dp_group_kernel --pairs 16384 \
   --note "NW l=1200 e=2% score-only, 16 384 pairs: dp_group_kernel with 20 registers (40 columns) per lane, 31 lanes per pair, two pairs per wavefront (round 6; dp_strip_kernel, one wavefront per pair, before)." \
   -- python3 tools/bench_configs.py nw_l1200_e2_score > $O/pmc_dpg20.log 2>&1; tail -1 $O/pmc_dpg20.log
python3 tools/pmc_summary.py --out $P/dp_group_l2000_cigar_pmc_summary.json --kernel dp_group_kernel --pairs 1024 \
   --note "SWG (int16 cells) l=2000 e=2% with CIGAR, 1 024 pairs: dp_group_kernel with ONE pair of 64 lanes per wavefront (round 6; dp_strip_kernel's two-wavefront strips before: 727 GCUPS)." \
   -- python3 tools/bench_configs.py swg_l2000_e2_w16_cigar > $O/pmc_dpg2000.log 2>&1; tail -1 $O/pmc_dpg2000.log
python3 tools/pmc_summary.py --out $P/dp_strip_nw_l5000_pmc_summary.json --kernel dp_strip_kernel --pairs 1024 \
   --note "NW l=5000 e=5% score-only (READ_SIZE 5264), 1 024 pairs: dp_strip_kernel, four wavefronts of 24 cells per lane per pair (round 6: until then the literal one-lane path behind an int16 bound that was loose by 2x: 4 GCUPS; NOTES R6.7)." \
   -- python3 tools/bench_configs.py nw_l5000_e5_score > $O/pmc_strip5000.log 2>&1; tail -1 $O/pmc_strip5000.log
SWEEP_NS=1024 SWEEP_LENGTHS=2000,2800,3500,5000,6200 python3 tools/strip_shape_sweep.py $P/strip_shape_sweep.txt > $O/strip_shape_sweep.log 2>&1        # (tests/test_strip_shape_cpu.py reads these two)
SWEEP_NS=256,512 SWEEP_LENGTHS=1000,2000,2800,3500,5000 python3 tools/strip_shape_sweep.py $P/strip_shape_sweep_few.txt > $O/strip_shape_sweep_few.log 2>&1
# (tools/cliff_scan.py -> $P/cliff_scan.txt takes ~11 minutes, most of it on the literal paths' rows: run it on its own)
python3 tools/length_sweep.py $P/length_sweep.txt > $O/length_sweep.log 2>&1
python3 tools/bench_configs.py > $P/all_configs_kernel_timers.jsonl 2> $O/configs.err
python3 -c "
import sys, json
for l in open('$P/all_configs_kernel_timers.jsonl'):
    d=json.loads(l); print('%-32s %-18s %10.4g pairs/s %8.1f GCUPS %8.1f GB/s' % (d['config'], d['kernel'], d['pairs_per_s'], d['gcups'], d['algorithmic_GBps']))
"
cp -r $P $O/
# the bench lines of the final code (one JSON line each; bench.py reads the PMC summaries written above)
python3 bench.py > $P/bench_default.json 2> $O/bench_default.err
python3 bench.py --backtrace --pairs 1048576 > $P/bench_cigar.json 2> $O/bench_cigar.err
for c in cfg3 cfg4 cfg5; do python3 bench.py --config $c --steps 3 --warmup 1 > $P/bench_$c.json 2> $O/bench_$c.err; done
cp -r $P $O/
