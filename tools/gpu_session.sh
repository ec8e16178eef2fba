#!/bin/bash
# One gpurun call: parity tests, headline bench, streaming bench, rocprofv3 kernel stats + PMC traffic passes.
# Everything lands in gpurun_out/.  Usage: tools/gpu_session.sh [tests|bench|prof|all ...]
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${*:-all}"
has() { [[ " $WHAT " == *" $1 "* || " $WHAT " == *" all "* ]]; }

if has tests; then
  (timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log
fi
if has bench; then
  (timeout 600 python bench.py --steps 3 --warmup 1 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_c3.json
  (timeout 300 python bench.py --workload c4_online --steps 12 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_c4.err | tail -1) > gpurun_out/bench_c4_graph.json
  (timeout 300 python bench.py --workload c4_online --steps 12 --warmup 3 --no-cpu-baseline --no-graph --no-profile 2>>gpurun_out/bench_c4.err | tail -1) > gpurun_out/bench_c4_nograph.json
  (timeout 300 python bench.py --workload c2_offline --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/bench_c2.err | tail -1) > gpurun_out/bench_c2.json
  (timeout 300 python bench.py --workload c3_offline --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_c3off.err | tail -1) > gpurun_out/bench_c3_offline.json
  (timeout 300 python bench.py --workload v2_sliding --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_v2.err | tail -1) > gpurun_out/bench_v2_sliding.json
  python - <<'PY'
import json
for f in ("bench_c3", "bench_c4_graph", "bench_c4_nograph", "bench_c2", "bench_c3_offline", "bench_v2_sliding"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("parity"), d.get("cpu_baseline", {}).get("value"))
        for k in d.get("kernels", []): print("   ", k)
        print("   roofline", {k: v for k, v in d.get("roofline", {}).items() if k != "note"})
    except Exception as e:
        print(f, "parse failed", e)
PY
  tail -5 gpurun_out/bench.err gpurun_out/bench_c4.err
fi
if has prof; then
  cd /tmp
  R=$GRAFT_REPO_ROOT
  CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile"
  (cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $CMD > $R/gpurun_out/prof_stats.log 2>&1)
  CMD0="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
  (cd $R && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- $CMD0 > $R/gpurun_out/prof_fetch.log 2>&1)
  (cd $R && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- $CMD0 > $R/gpurun_out/prof_write.log 2>&1)
  cd $R
  python tools/summarize_rocprof.py gpurun_out/prof_stats gpurun_out/rocprof_kernel_stats.txt | head -30
  F=$(ls -t $(find gpurun_out/prof_fetch -name "*counter_collection.csv") | head -1)
  W=$(ls -t $(find gpurun_out/prof_write -name "*counter_collection.csv") | head -1)
  python tools/pmc_traffic.py "$F" "$W" gpurun_out/pmc_traffic.json
  # keep only the small summaries (raw traces are large)
  find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name '*kernel_trace.csv' -delete
  du -sh gpurun_out
fi
