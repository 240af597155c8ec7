#!/bin/bash
# tools/ab_configs.sh rounds "cfg1 cfg2 ..." name... : interleaved rounds of tools/bench_configs.py over build_ab/lib_<name>.so (GPU box)
rounds=$1; cfgs=$2; shift 2
for r in $(seq $rounds); do for v in "$@"; do
  AIM_LIB=$PWD/build_ab/lib_$v.so timeout 600 python tools/bench_configs.py $cfgs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['config'], round(d['kernel_ms'], 4), '%.4g' % d['pairs_per_s'])"
done; done
