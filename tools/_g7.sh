mkdir -p gpurun_out/r05g
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nw_ or swg or dp_lane or sample or judge or edge or synthetic or prefilled" 2>&1 | tail -3
python3 tools/bench_configs.py nw_l100_e1_cigar nw_l100_e5_cigar nw_l100_e10_cigar swg_l100_e1_cigar swg_l100_e5_cigar > gpurun_out/r05g/b.jsonl 2>gpurun_out/r05g/err
python3 -c "
import json
for l in open('gpurun_out/r05g/b.jsonl'):
    d=json.loads(l); print(d['config'], d['kernel'], '%.3f ms'%d['kernel_ms'], '%.0f GCUPS'%d['gcups'], 'todo', d.get('todo_pairs'))"
AIM_DEBUG_POISON_OPS=170 timeout 400 python3 tools/fuzz_parity.py --focus dplane --seconds 300 > gpurun_out/r05g/fuzz_dplane.log 2>&1; tail -2 gpurun_out/r05g/fuzz_dplane.log | cut -c1-300
