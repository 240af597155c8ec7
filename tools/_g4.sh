mkdir -p gpurun_out/r05d
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dp_wave or cfg4 or judge or dp_lane_all" 2>&1 | tail -3
python3 tools/bench_configs.py swg_l10000_e1_cigar_n256 swg_l1000_e5_cigar > gpurun_out/r05d/bench.jsonl 2>gpurun_out/r05d/bench.err
AIM_DEBUG_FLAGS=1 python3 tools/bench_configs.py swg_l10000_e1_cigar_n256 swg_l1000_e5_cigar >> gpurun_out/r05d/bench.jsonl 2>>gpurun_out/r05d/bench.err
python3 -c "
import json
for l in open('gpurun_out/r05d/bench.jsonl'):
    d=json.loads(l); print(d['config'], d['kernel'], '%.3f ms'%d['kernel_ms'], '%.0f GCUPS'%d['gcups'], 'todo', d.get('todo_pairs'))"
timeout 300 python3 tools/fuzz_parity.py --focus dp --seconds 200 > gpurun_out/r05d/fuzz_dp.log 2>&1; tail -1 gpurun_out/r05d/fuzz_dp.log | cut -c1-300
