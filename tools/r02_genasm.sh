#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ga
timeout 900 python -m pytest tests/test_genasm.py tests/test_gpu_parity.py -x -q -m gpu -k "genasm" 2>&1 | tail -3 > gpurun_out/ga/pytest.txt
timeout 300 python tools/fuzz_parity.py --seconds 150 --seed 777 --focus genasm 2>&1 | tail -2 > gpurun_out/ga/fuzz.txt
python tools/bench_configs.py genasm_l100000_e10_cigar genasm_l100000_e10_score genasm_l100000_e10_cigar_n4096 genasm_l10000_e10_cigar genasm_l100_e10_cigar 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], round(d['kernel_ms'], 4), '%.4g' % d['pairs_per_s'])" > gpurun_out/ga/bench.txt
cat gpurun_out/ga/pytest.txt gpurun_out/ga/fuzz.txt gpurun_out/ga/bench.txt
AIM_LIB=$PWD/build_ab/lib_gastamps.so python tools/ga_stamps.py 100000 0.10 1024 2>&1 | tail -8
