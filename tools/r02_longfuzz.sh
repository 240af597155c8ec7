#!/bin/bash
# long differential runs on the final code of the round (kernel level, CLI level, dp_wave soak)
cd $GRAFT_REPO_ROOT; O=gpurun_out/lf; mkdir -p $O
timeout 700 python tools/fuzz_parity.py --seconds 600 --seed 8101 > $O/fuzz_all.txt 2>&1; tail -1 $O/fuzz_all.txt
timeout 400 python tools/fuzz_parity.py --seconds 300 --seed 8102 --focus genasm > $O/fuzz_genasm.txt 2>&1; tail -1 $O/fuzz_genasm.txt
timeout 400 python tools/fuzz_parity.py --seconds 300 --seed 8103 --focus wfa > $O/fuzz_wfa.txt 2>&1; tail -1 $O/fuzz_wfa.txt
timeout 500 python tools/fuzz_cli.py --seconds 400 --seed 8104 > $O/fuzz_cli.txt 2>&1; tail -1 $O/fuzz_cli.txt
timeout 400 python tools/soak_dp_wave.py --seconds 250 --slots 8 --seed 8105 > $O/soak.json 2> $O/soak.err; echo "soak rc=$?"
