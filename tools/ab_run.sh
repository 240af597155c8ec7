#!/bin/bash
# tools/ab_run.sh rounds name...  : interleaved rounds of bench.py over build_ab/lib_<name>.so (run on the GPU box)
rounds=$1; shift
for r in $(seq $rounds); do for v in "$@"; do
  AIM_LIB=$PWD/build_ab/lib_$v.so timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['kernel_ms'],4), round(d['roofline']['frac'],4), d['verified_vs_oracle'])"
done; done
