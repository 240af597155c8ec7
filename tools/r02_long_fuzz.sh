#!/bin/bash
# long differential runs with the GPU minutes that are left (all against the oracle, bit-exact)
O=gpurun_out/r02h; mkdir -p $O
timeout 1000 python tools/fuzz_parity.py --seconds 900 --seed 71 > $O/fuzz_parity_all.log 2>&1; tail -1 $O/fuzz_parity_all.log | cut -c1-300
timeout 500 python tools/fuzz_parity.py --seconds 400 --seed 72 --focus lane > $O/fuzz_parity_lane.log 2>&1; tail -1 $O/fuzz_parity_lane.log | cut -c1-300
timeout 500 python tools/fuzz_parity.py --seconds 400 --seed 73 --focus genasm > $O/fuzz_parity_genasm.log 2>&1; tail -2 $O/fuzz_parity_genasm.log | cut -c1-300
timeout 800 python tools/fuzz_cli.py --seconds 700 --seed 74 > $O/fuzz_cli.log 2>&1; tail -2 $O/fuzz_cli.log | cut -c1-300
timeout 500 python tools/fuzz_cli.py --seconds 400 --seed 75 --focus dplong > $O/fuzz_cli_dplong.log 2>&1; tail -2 $O/fuzz_cli_dplong.log | cut -c1-300
timeout 700 python tools/soak_dp_wave.py --seconds 600 --slots 8 --seed 9 > $O/soak.json 2> $O/soak.err; python -c "
import json; d=json.load(open('$O/soak.json')); print('soak', d['launches'], d['failed_slot'])"
