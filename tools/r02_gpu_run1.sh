#!/bin/bash
# round-2 GPU run 1: parity suite, soak (plain + LDS poison), headline bench
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout 600 python tools/soak_dp_wave.py --seconds 300 --slots 8 > gpurun_out/r02a/soak_plain.json 2> gpurun_out/r02a/soak_plain.err; echo "soak rc=$?"
timeout 300 python tools/soak_dp_wave.py --seconds 120 --slots 8 --poison-lds 255 --seed 2 > gpurun_out/r02a/soak_ff.json 2> gpurun_out/r02a/soak_ff.err; echo "soak ff rc=$?"
timeout 300 python tools/soak_dp_wave.py --seconds 120 --slots 8 --poison-lds 0 --seed 3 > gpurun_out/r02a/soak_00.json 2> gpurun_out/r02a/soak_00.err; echo "soak 00 rc=$?"
python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
cat gpurun_out/r02a/bench.json | head -c 1500
python -c "
import json
for f in ('soak_plain','soak_ff','soak_00'):
    try:
        d=json.load(open('gpurun_out/r02a/%s.json'%f)); print(f, d['launches'], d['failed_slot'], d['seconds'])
    except Exception as e: print(f, 'ERR', e)
"
