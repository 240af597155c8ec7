#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/dpw; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dp or nw or swg or judge or golden or digest or cli" 2>&1 | tail -3 > $O/pytest.txt
timeout 300 python tools/fuzz_parity.py --seconds 200 --seed 6601 > $O/fuzz.txt 2>&1
bash tools/ab_configs.sh 2 "swg_l1000_e5_score nw_l1000_e5_score swg_l10000_e1_score_n256 swg_l1000_e5_cigar nw_l1000_e5_cigar swg_l10000_e1_cigar_n256" dbase dnew > $O/ab.txt 2>&1
cat $O/pytest.txt; tail -1 $O/fuzz.txt; sort -k2,2 -k1,1 -s $O/ab.txt
