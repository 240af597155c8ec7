#!/bin/bash
# l=100 e=5% score-only on wfa_group_kernel: phase stamps (stamped build) and PMC counters (production build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e5
AIM_LIB=$PWD/build_ab/lib_gstamps.so python tools/group_stamps.py auto 100 0.05 1048576 > gpurun_out/e5/stamps.txt 2>&1
AIM_LIB=$PWD/build_ab/lib_gstamps.so python tools/group_stamps.py auto 100 0.10 524288 >> gpurun_out/e5/stamps.txt 2>&1
python3 tools/pmc_summary.py --out gpurun_out/e5/wfa_group_l100_e5_pmc_summary.json --kernel wfa_group_kernel --pairs 1048576 -- python3 tools/bench_configs.py wfa_l100_e5_score > gpurun_out/e5/pmc.log 2>&1
cat gpurun_out/e5/stamps.txt
