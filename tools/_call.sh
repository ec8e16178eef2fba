mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x -k "attention or update_former or forward_window" 2>&1 | tail -4) > gpurun_out/pytest_attn.log; tail -3 gpurun_out/pytest_attn.log
(timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_c3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_c3.json"))
print("bench_c3", d["value"], d["ms_per_step"], d.get("parity"))
for k in d.get("kernels", [])[:10]: print("   ", k["name"], k["launches"], k["total_ms"], k["avg_us"])
PY
