#!/bin/bash
# last evidence pass of round 2: all-config kernel timers on the final plans, full GPU suite, one more fuzz / soak leg
cd $GRAFT_REPO_ROOT; O=gpurun_out/r02x; mkdir -p $O
python3 tools/bench_configs.py > $O/all_configs_kernel_timers.jsonl 2> $O/configs.err
python3 -c "
import sys, json
for l in open('$O/all_configs_kernel_timers.jsonl'):
    d=json.loads(l); print('%-32s %-18s %10.4g pairs/s %8.1f GCUPS' % (d['config'], d['kernel'], d['pairs_per_s'], d['gcups']))
"
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 500 python tools/fuzz_parity.py --seconds 400 --seed 9901 > $O/fuzz_all.txt 2>&1; tail -1 $O/fuzz_all.txt
timeout 300 python tools/fuzz_cli.py --seconds 200 --seed 9902 > $O/fuzz_cli.txt 2>&1; tail -1 $O/fuzz_cli.txt
timeout 300 python tools/soak_dp_wave.py --seconds 150 --slots 8 --seed 9903 > $O/soak.json 2> $O/soak.err; echo "soak rc=$?"
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['roofline']['frac'], d['verified_vs_oracle'])"
