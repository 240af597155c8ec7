mkdir -p gpurun_out/r05e
python -m pytest tests/ -x -q -m gpu > gpurun_out/r05e/pytest_all.log 2>&1; tail -12 gpurun_out/r05e/pytest_all.log
python3 tools/bench_configs.py wfa_l100_e1_x4g6a2_score wfa_l100_e1_x4g6a2_cigar wfa_l100_e1_score wfa_l100_e1_cigar swg_l10000_e1_cigar_n256 > gpurun_out/r05e/bench.jsonl 2>gpurun_out/r05e/bench.err
AIM_NO_LANE=1 python3 tools/bench_configs.py wfa_l100_e1_x4g6a2_score wfa_l100_e1_x4g6a2_cigar >> gpurun_out/r05e/bench.jsonl 2>>gpurun_out/r05e/bench.err
for k in 16 24 32; do AIM_STRIP_K=$k python3 tools/bench_configs.py swg_l10000_e1_cigar_n256 >> gpurun_out/r05e/bench.jsonl 2>>gpurun_out/r05e/bench.err; done
python3 -c "
import json
for l in open('gpurun_out/r05e/bench.jsonl'):
    d=json.loads(l); print(d['config'], d['kernel'], '%.3f ms'%d['kernel_ms'], '%.3g pairs/s'%d['pairs_per_s'], d['plan'][:60])"
python3 bench.py --backtrace --pairs 1048576 --no-cpu-baseline --no-e2e > gpurun_out/r05e/bench_cigar.json 2>gpurun_out/r05e/bench_cigar.err; cut -c1-400 gpurun_out/r05e/bench_cigar.json
