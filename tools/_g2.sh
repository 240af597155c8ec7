mkdir -p gpurun_out/r05b
python3 tools/bench_configs.py swg_l100_e1_score swg_l100_e2_score swg_l100_e5_score swg_l100_e5_cigar swg_l100_e10_score swg_l70_e2_score > gpurun_out/r05b/bench2.jsonl 2>gpurun_out/r05b/bench2.err; cut -c1-330 gpurun_out/r05b/bench2.jsonl
AIM_NO_SWG_REG=1 python3 tools/bench_configs.py swg_l100_e5_score swg_l100_e10_score swg_l70_e2_score > gpurun_out/r05b/bench2_lane.jsonl 2>>gpurun_out/r05b/bench2.err; cut -c1-330 gpurun_out/r05b/bench2_lane.jsonl
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out gpurun_out/r05b/swg_reg_pmc_summary.json --kernel swg_reg_kernel --pairs 1048576 --note "swg_reg first version: SWG l=100 e=1% score-only, 1 Mi pairs" -- python3 tools/bench_configs.py swg_l100_e1_score > gpurun_out/r05b/pmc.log 2>&1; tail -1 gpurun_out/r05b/pmc.log
timeout 400 python3 tools/fuzz_parity.py --focus dplane --seconds 300 > gpurun_out/r05b/fuzz_dplane.log 2>&1; tail -3 gpurun_out/r05b/fuzz_dplane.log
python -m pytest tests/ -x -q -m gpu > gpurun_out/r05b/pytest_all.log 2>&1; tail -5 gpurun_out/r05b/pytest_all.log
