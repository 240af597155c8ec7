#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('%.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle']); print(json.dumps(d['e2e'], indent=1)); print(d['cpu_baseline'])
"
python bench.py --backtrace --pairs 1048576 --no-cpu-baseline > $O/bench_cigar.json 2> $O/bench_cigar.err
python -c "
import json
d=json.loads(open('$O/bench_cigar.json').read().strip().splitlines()[-1]); print('cigar %.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle']); print(json.dumps(d['e2e'], indent=1))
"
