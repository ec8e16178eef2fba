#!/bin/bash
# One gpurun call: parity tests, GEMM micro-bench, headline bench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
(ROUNDS=5 timeout 300 python tools/bench_gemm.py 2>&1 | tail -60) > gpurun_out/bench_gemm.log
(timeout 600 python bench.py --steps 2 --warmup 1 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_f16x3.json
(timeout 600 python bench.py --steps 2 --warmup 1 --precision f32 --no-cpu-baseline 2>>gpurun_out/bench.err | tail -1) > gpurun_out/bench_f32.json
tail -5 gpurun_out/pytest_gpu.log; tail -22 gpurun_out/bench_gemm.log; head -c 1500 gpurun_out/bench_f16x3.json
