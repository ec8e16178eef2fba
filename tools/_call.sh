mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q -x -k "gemm or update_former or forward_window" 2>&1 | tail -4) > gpurun_out/pytest_gemm.log; tail -3 gpurun_out/pytest_gemm.log
for t in 0 1; do
  (CTK_GEMM_EPI=$t MODES=sh,sh2sh ROUNDS=4 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/bench_gemm_epi$t.txt
  echo "== epi $t"; grep -v "^shape" gpurun_out/bench_gemm_epi$t.txt | awk '{print $1, $5, $6}' | tr '\n' ';'; echo
done
(timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_c3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_c3.json"))
print("bench_c3", d["value"], d["ms_per_step"], d.get("parity"))
for k in d.get("kernels", [])[:8]: print("   ", k)
PY
