#!/bin/bash
# persistent-GEMM A/B: production shapes of one C3 iteration (tools/bench_gemm.py, SH in -> f32 / SH out) per knob setting
mkdir -p gpurun_out
for cfg in "0 0" "1 0" "2 0" "1 1"; do
  set -- $cfg
  echo "=== CTK_GEMM_PERSIST=$1 CTK_GEMM_TILE=$2"
  CTK_GEMM_PERSIST=$1 CTK_GEMM_TILE=$2 MODES=sh,sh2sh ROUNDS=4 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r02_gemm_persist_ab.txt 2>&1
cat gpurun_out/r02_gemm_persist_ab.txt
CTK_GEMM_PERSIST=1 timeout 200 python tools/bench_gemm_sweep.py 2>&1 | tail -30 > gpurun_out/r02_gemm_sweep_persist.txt
tail -4 gpurun_out/r02_gemm_sweep_persist.txt
# correctness of the persistent path on the GEMM / model tests
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gemm or update_former or forward_window or full_size" 2>&1 | tail -5
