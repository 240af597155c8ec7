#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/pc; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wfa or group or judge or golden or digest" 2>&1 | tail -2 > $O/pytest.txt
bash tools/ab_configs.sh 3 "wfa_l100_e10_score wfa_l1000_e5_score wfa_l150_e2_score wfa_l1000_e5_cigar wfa_l250_e5_score" pbase pnew > $O/ab.txt 2>&1
cat $O/pytest.txt; sort -k2,2 -k1,1 -s $O/ab.txt
