mkdir -p gpurun_out
for t in 0 1 2 3; do
  (CTK_GEMM_STAGGER=$t MODES=sh,sh2sh ROUNDS=4 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/bench_gemm_stagger$t.txt
  echo "== stagger $t"; grep -v "^shape" gpurun_out/bench_gemm_stagger$t.txt | awk '{print $1, $5, $6}' | tr '\n' ';'; echo
done
