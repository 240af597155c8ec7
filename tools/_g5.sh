mkdir -p gpurun_out/r05e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "penalty_sets or prefilled or lane or fused or sample" > gpurun_out/r05e/pytest.log 2>&1; tail -5 gpurun_out/r05e/pytest.log
export AIM_DEBUG_POISON_OPS=238
for f in lane wfa fused; do timeout 200 python3 tools/fuzz_parity.py --focus $f --seconds 100 > gpurun_out/r05e/fuzz_$f.log 2>&1; tail -2 gpurun_out/r05e/fuzz_$f.log | cut -c1-250; done
timeout 200 python3 tools/fuzz_parity.py --focus dplane --seconds 100 > gpurun_out/r05e/fuzz_dplane.log 2>&1; tail -2 gpurun_out/r05e/fuzz_dplane.log | cut -c1-250
unset AIM_DEBUG_POISON_OPS
python3 tools/bench_configs.py wfa_l100_e1_x4g6a2_score wfa_l100_e1_x4g6a2_cigar > gpurun_out/r05e/bench.jsonl 2>gpurun_out/r05e/bench.err
AIM_NO_LANE=1 python3 tools/bench_configs.py wfa_l100_e1_x4g6a2_score wfa_l100_e1_x4g6a2_cigar >> gpurun_out/r05e/bench.jsonl 2>>gpurun_out/r05e/bench.err
python3 -c "
import json
for l in open('gpurun_out/r05e/bench.jsonl'):
    d=json.loads(l); print(d['config'], d['kernel'], '%.3f ms'%d['kernel_ms'], '%.3g pairs/s'%d['pairs_per_s'], d['plan'][:60])"
python3 bench.py --backtrace --pairs 1048576 --no-cpu-baseline --no-e2e > gpurun_out/r05e/bench_cigar.json 2>gpurun_out/r05e/bench_cigar.err; cut -c1-200 gpurun_out/r05e/bench_cigar.json
