#!/bin/bash
# Round-2 GPU session: tools/gpu_session_r02.sh [tests] [bench] [c5] [enc] [prof] [pmc]
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${*:-tests bench}"
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  (timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60) > gpurun_out/r02_pytest_gpu.log
  tail -25 gpurun_out/r02_pytest_gpu.log
fi
if has bench; then
  (timeout 900 python bench.py --steps 3 --warmup 1 2>gpurun_out/r02_bench.err | tail -1) > gpurun_out/r02_bench_c3.json
  python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r02_bench_c3.json"))
    print("c3", d["value"], d["ms_per_step"], json.dumps(d.get("parity")), json.dumps(d.get("cpu_baseline")))
    for k in d.get("kernels", []): print("   ", k)
    for key in ("roofline", "roofline_gemm", "roofline_sampler"):
        print(key, {k: v for k, v in d.get(key, {}).items() if k not in ("note", "traffic_detail", "traffic_note")})
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r02_bench.err").read()[-3000:])
PY
fi
if has c5; then
  (timeout 600 python bench.py --workload c5_shard --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02_bench_c5.err | tail -1) > gpurun_out/r02_bench_c5.json
  (timeout 300 python bench.py --workload c2_offline --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/r02_bench_c2.err | tail -1) > gpurun_out/r02_bench_c2.json
  (timeout 300 python bench.py --workload c4_online --steps 12 --warmup 3 --no-cpu-baseline 2>gpurun_out/r02_bench_c4.err | tail -1) > gpurun_out/r02_bench_c4.json
  (timeout 300 python bench.py --workload v2_sliding --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/r02_bench_v2.err | tail -1) > gpurun_out/r02_bench_v2.json
  python - <<'PY'
import json
for f in ("c5", "c2", "c4", "v2"):
    try:
        d = json.load(open(f"gpurun_out/r02_bench_{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["config"].get("points_per_gpu"), json.dumps(d.get("parity"))[:400])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/r02_bench_{f}.err").read()[-1500:])
PY
fi
if has dist; then
  # the multi-rank code path on a 1-GPU box: 2 ranks over gloo, both on cuda:0 (numbers meaningless, path exercised)
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --workload tiny --dist-backend gloo --single-device 2>gpurun_out/r02_bench_2rank.err | tail -1) > gpurun_out/r02_bench_2rank_gloo.json
  (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --workload c5_shard --dist-backend gloo --single-device --no-profile 2>>gpurun_out/r02_bench_2rank.err | tail -1) > gpurun_out/r02_bench_2rank_c5_gloo.json
  head -c 600 gpurun_out/r02_bench_2rank_gloo.json; echo; head -c 600 gpurun_out/r02_bench_2rank_c5_gloo.json; echo; tail -3 gpurun_out/r02_bench_2rank.err
fi
if has enc; then
  (timeout 600 python tools/probe_encoder_precision.py 2>&1 | tail -8) > gpurun_out/r02_encoder_precision.txt
  cat gpurun_out/r02_encoder_precision.txt
fi
if has prof; then
  R=$GRAFT_REPO_ROOT
  cd /tmp
  CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile"
  (cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $CMD > $R/gpurun_out/prof_stats.log 2>&1)
  cd $R
  python tools/summarize_rocprof.py gpurun_out/prof_stats gpurun_out/r02_rocprof_kernel_stats.txt | head -40
  find gpurun_out/prof_stats -name '*kernel_trace.csv' -delete
fi
if has pmc; then
  R=$GRAFT_REPO_ROOT
  cd /tmp
  CMD0="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
  (cd $R && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- $CMD0 > $R/gpurun_out/prof_fetch.log 2>&1)
  (cd $R && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- $CMD0 > $R/gpurun_out/prof_write.log 2>&1)
  cd $R
  F=$(ls -t $(find gpurun_out/prof_fetch -name "*counter_collection.csv") | head -1)
  W=$(ls -t $(find gpurun_out/prof_write -name "*counter_collection.csv") | head -1)
  python tools/pmc_traffic.py "$F" "$W" gpurun_out/r02_pmc_traffic.json
  find gpurun_out/prof_fetch gpurun_out/prof_write -name '*.csv' -size +20M -delete
  du -sh gpurun_out
fi
if has sq; then
  # SQ counters per kernel (MFMA-pipe utilisation, issue / wait split) and an LDS pass, one bench step each
  R=$GRAFT_REPO_ROOT
  cd /tmp
  CMD0="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile"
  (cd $R && timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_sq -- $CMD0 > $R/gpurun_out/prof_sq.log 2>&1)
  (cd $R && timeout 900 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_lds -- $CMD0 > $R/gpurun_out/prof_lds.log 2>&1)
  cd $R
  F=$(ls -t $(find gpurun_out/prof_sq -name "*counter_collection.csv") | head -1)
  python tools/summarize_sq.py "$F" "" gpurun_out/r02_sq_counters.txt | cut -c1-260
  F=$(ls -t $(find gpurun_out/prof_lds -name "*counter_collection.csv") | head -1)
  python tools/summarize_lds.py "$F" gpurun_out/r02_lds_counters.txt | cut -c1-220
  find gpurun_out/prof_sq gpurun_out/prof_lds -name '*.csv' -size +20M -delete
fi
