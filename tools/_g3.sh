mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swg or dp_lane or sample_file or edge or judge or synthetic_l100 or dp_wave or cfg4" > gpurun_out/r05c/pytest.log 2>&1; tail -15 gpurun_out/r05c/pytest.log
python3 tools/bench_configs.py swg_l100_e1_score swg_l100_e2_score swg_l100_e5_score swg_l100_e1_cigar swg_l100_e5_cigar swg_l70_e2_score > gpurun_out/r05c/bench.jsonl 2>gpurun_out/r05c/bench.err; python3 -c "
import json
for l in open('gpurun_out/r05c/bench.jsonl'):
    d=json.loads(l); print(d['config'], d['kernel'], '%.3f ms'%d['kernel_ms'], '%.0f GCUPS'%d['gcups'], 'todo', d.get('todo_pairs'))"
tail -3 gpurun_out/r05c/bench.err
timeout 300 python3 tools/fuzz_parity.py --focus dplane --seconds 200 > gpurun_out/r05c/fuzz_dplane.log 2>&1; tail -3 gpurun_out/r05c/fuzz_dplane.log
