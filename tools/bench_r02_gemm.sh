#!/bin/bash
# full-row (128 x 384, 8 waves) tile for the N = 384 Linears vs the default 128 x 128: production shapes + in situ
mkdir -p gpurun_out
for t in 0 4; do
  echo "=== CTK_GEMM_TILE=$t"
  CTK_GEMM_TILE=$t MODES=sh ROUNDS=4 SHAPES=corr_fc1,in_proj,q_all,out_all,fc2_all,q_pts,out_pts,fc2_pts timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r02_gemm_fullrow_ab.txt 2>&1
cat gpurun_out/r02_gemm_fullrow_ab.txt
for t in 0 7 4; do
  (CTK_GEMM_TILE=$t timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r02_var_tile$t.json
  python - $t <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r02_var_tile{sys.argv[1]}.json"))
print("CTK_GEMM_TILE", sys.argv[1], d["value"], d["ms_per_step"], [(r["name"], r["avg_us"], r["total_ms"]) for r in d["kernels"] if "n384" in r["name"]])
PY
done
