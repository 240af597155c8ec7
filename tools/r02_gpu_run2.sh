#!/bin/bash
# round-2 GPU run 2: new tests (lane raw path, compact IO, slots/packed/CIGAR), bench both layouts, PMC summary of wfa_lane
O=gpurun_out/r02b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for io in compact default; do python bench.py --io $io > $O/bench_$io.json 2> $O/bench_$io.err; echo "bench $io rc=$?"; done
python bench.py --io compact > $O/bench_compact2.json 2>> $O/bench_compact.err
python bench.py --backtrace --pairs 1048576 > $O/bench_cigar.json 2> $O/bench_cigar.err; echo "bench cigar rc=$?"
for f in compact default compact2 cigar; do python -c "
import json,sys
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', '%.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle'])
"; done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/pmc_summary.py --out profiles/r02/wfa_lane_pmc_summary.json --kernel wfa_lane_kernel --pairs 4194304 --alg-bytes 905968812 --fetch-x2 --io compact \
   --note "Cross-check: 4194304 pairs x (224 B rows + 8 B request) = 973.1 MB read, x 8 B result = 33.6 MB written." \
   -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_lane.log 2>&1; tail -2 $O/pmc_lane.log
cp profiles/r02/*.json profiles/r02/*.csv $O/ 2>/dev/null
