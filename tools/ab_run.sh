#!/bin/bash
# A/B on one box: the listed bench_configs rows with build_ab/lib_<name>.so (default base) and with the working tree's library, alternating, twice.
# tools/ab_run.sh <name> <config>...
cd "$(dirname "$0")/.."
name=$1; shift
for rep in 1 2; do
  for lib in build_ab/lib_$name.so aim_amd/libaim_hip.so; do
    AIM_LIB=$PWD/$lib python tools/bench_configs.py "$@" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-28s %-34s %9.4f ms  %s' % ('$lib'.split('/')[-1], d['config'], d.get('kernel_ms', -1), d.get('error', '')))"
  done
done
