mkdir -p gpurun_out/r05f
python3 tools/bench_configs.py nw_l100_e1_score nw_l100_e10_score > gpurun_out/r05f/bench.jsonl 2>gpurun_out/r05f/bench.err
AIM_DEBUG_FLAGS=2 python3 tools/bench_configs.py nw_l100_e10_score >> gpurun_out/r05f/bench.jsonl 2>>gpurun_out/r05f/bench.err
python3 -c "
import json
for l in open('gpurun_out/r05f/bench.jsonl'):
    d=json.loads(l); print(d['config'], d['kernel'], '%.3f ms'%d['kernel_ms'], '%.0f GCUPS'%d['gcups'], 'todo', d.get('todo_pairs'))"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nw_row or nw_asym" 2>&1 | tail -2
