#!/bin/bash
# Round-2 A/B runs on the C3 workload (one gpurun call): two-stream overlap modes and the persistent time attention.
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  (env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02_var_$name.err | tail -1) > gpurun_out/r02_var_$name.json
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r02_var_{n}.json"))
    k = {r["name"]: (r["avg_us"], r["total_ms"]) for r in d.get("kernels", [])}
    print(n, d["value"], d["ms_per_step"], "| corr", k.get("corr_volume_sh"), "fc1", k.get("gemm_sh_128_k2432_n384"),
          "attn_time", k.get("attention_time"), "qout", k.get("gemm_sh_128_k384_n384"), "parity", d["parity"].get("c2", {}).get("coords_px"))
except Exception as e:
    print(n, "failed", e); print(open(f"gpurun_out/r02_var_{n}.err").read()[-1500:])
PY
}
run base CTK_OVERLAP=0 CTK_ATTN_TIME=0
run attn CTK_OVERLAP=0
run ov1 CTK_OVERLAP=1
run ov2 CTK_OVERLAP=2
run ov3 CTK_OVERLAP=3
