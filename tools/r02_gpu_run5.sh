#!/bin/bash
O=gpurun_out/r02e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python tools/bench_configs.py > $O/all_configs_kernel_timers.jsonl 2> $O/configs.err; python3 -c "
import sys, json
for l in open('$O/all_configs_kernel_timers.jsonl'):
    d=json.loads(l); print('%-28s %-18s %10.4g pairs/s %8.1f GCUPS %8.1f GB/s' % (d['config'], d['kernel'], d['pairs_per_s'], d['gcups'], d['algorithmic_GBps']))
"
timeout 300 python tools/fuzz_parity.py --seconds 200 --seed 51 | tail -1 | cut -c1-300
