#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sample or lane or cfg2 or compact or synthetic_l100 or fast_path or ragged or edge or real_reads or judge or packed or slots or host_cli" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do for io in compact default; do python bench.py --io $io --no-cpu-baseline > $O/bench_${io}_$i.json 2> $O/bench_${io}_$i.err; python -c "
import json
d=json.loads(open('$O/bench_${io}_$i.json').read().strip().splitlines()[-1]); print('$io', '%.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle'])
"; done; done
python bench.py --backtrace --pairs 1048576 --no-cpu-baseline > $O/bench_cigar.json 2> $O/bench_cigar.err; python -c "
import json
d=json.loads(open('$O/bench_cigar.json').read().strip().splitlines()[-1]); print('cigar', '%.4g pairs/s'%d['value'], 'kernel_ms %.4f'%d['kernel_ms'], 'frac %.4f'%d['roofline']['frac'], d['verified_vs_oracle'])
"
