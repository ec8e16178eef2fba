#!/bin/bash
# One gpurun call: parity tests, headline bench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "corr_volume_sh" 2>&1 | tail -40) > gpurun_out/pytest_corr.log
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_t1.json
tail -30 gpurun_out/pytest_corr.log; tail -12 gpurun_out/pytest_gpu.log; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_t1.json'))
    print(d['value'], d['ms_per_step'], d['parity'])
    for k in d['kernels']: print('   ', k)
except Exception as e: print('bench parse failed', e); print(open('gpurun_out/bench.err').read()[-2000:])
PY
