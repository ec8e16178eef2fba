#!/bin/bash
# one-wave-per-SIMD tiles for the N = 384 Linears (CTK_GEMM_TILE=8: 128x384 / 4 waves / 2 stages; 9: 256x128 / 4 waves / 3 stages)
mkdir -p gpurun_out
for t in 8 9; do
  CTK_GEMM_TILE=$t timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_gemm or update_former or test_corr_embed" 2>&1 | tail -2
done
for t in 0 8 9; do
  echo "=== CTK_GEMM_TILE=$t"
  CTK_GEMM_TILE=$t MODES=sh ROUNDS=4 SHAPES=corr_fc1,in_proj,q_all,out_all,fc2_all,q_pts,out_pts,fc2_pts timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r02_gemm_onewave_ab.txt 2>&1
cat gpurun_out/r02_gemm_onewave_ab.txt
