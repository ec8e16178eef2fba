mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
(timeout 600 python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_c3.json
(timeout 300 python bench.py --workload c4_online --steps 12 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_c4.err | tail -1) > gpurun_out/bench_c4_graph.json
python - <<'PY'
import json
for f in ("bench_c3", "bench_c4_graph"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("parity"))
        for k in d.get("kernels", []): print("   ", k)
    except Exception as e:
        print(f, "parse failed", e)
PY
