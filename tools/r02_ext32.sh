#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ext32
AIM_LIB=$PWD/build_ab/lib_rskip.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "wfa or group or judge or golden or digest" 2>&1 | tail -3 > gpurun_out/ext32/pytest.txt
AIM_LIB=$PWD/build_ab/lib_rskip.so timeout 400 python tools/fuzz_parity.py --seconds 200 --seed 9102 --focus wfa 2>&1 | tail -1 > gpurun_out/ext32/fuzz.txt
bash tools/ab_configs.sh 3 "wfa_l100_e5_score wfa_l100_e10_score wfa_l150_e2_score wfa_l250_e5_score wfa_l100_e5_cigar wfa_l1000_e5_cigar wfa_l1000_e5_score" rbase rskip > gpurun_out/ext32/ab.txt 2>&1
cat gpurun_out/ext32/pytest.txt gpurun_out/ext32/fuzz.txt; sort -k2,2 -k1,1 -s gpurun_out/ext32/ab.txt
