#!/bin/bash
# Round-6 GPU session: tools/gpu_session.sh [pmc] [tests] [bench] [more] [prof] [sq] [lab] [retest] [soak]   (default: tests bench)
# One gpurun call = one box: everything wanted from it is listed here; outputs go to gpurun_out/r06_*.
# The LAST call of a round runs "pmc tests bench more prof sq" on the final tree: profiles/pmc_traffic.json is stamped with the
# hash of the kernel sources (__graft_entry__.source_hash), so any later edit under csrc/ makes it stale for bench.py.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r06
SRC_HASH=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
echo "kernel sources sha256 $SRC_HASH"
WHAT="${*:-tests bench}"
has() { [[ " $WHAT " == *" $1 "* ]]; }
summ() {  # summ <tag> <json>
  python - "$1" "$2" <<'PY'
import json, sys
tag, f = sys.argv[1:3]
try:
    d = json.load(open(f))
    print(tag, d["value"], d["ms_per_step"], "ms; parity:", json.dumps(d.get("parity"))[:600])
    if tag == "c3":
        for k in d.get("kernels", [])[:28]: print("   ", k)
        for key in ("roofline", "roofline_gemm", "roofline_sampler", "sustained_mfma", "extra_lines", "cpu_baseline"):
            v = d.get(key) or {}
            print(key, {k: x for k, x in v.items() if k not in ("note", "traffic_detail", "traffic_note", "reference_in_build_container", "sample", "rows")})
        for r in (d.get("roofline_gemm") or {}).get("rows", []): print("      ", r)
except Exception as e:
    print(tag, "parse failed", e)
    try: print(open(f.replace(".json", ".err")).read()[-3000:])
    except Exception: pass
PY
}
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    cd /tmp && rm -rf /tmp/pmc_$c && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-lines > /tmp/pmc_$c.log 2>&1
    cd $GRAFT_REPO_ROOT
  done
  ff=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$ff" "$fw" gpurun_out/${R}_pmc_traffic.json c3_sliding > gpurun_out/${R}_pmc_traffic.txt 2>&1
  head -30 gpurun_out/${R}_pmc_traffic.txt
  cp gpurun_out/${R}_pmc_traffic.json profiles/pmc_traffic.json  # (on the box: the bench lines below attach it; its library hash is this build's)
fi
if has lab; then  # GEMM lab (linked against the DEV library): fp64 / determinism / timing of every Linear shape, the shader clock under load
  (timeout 600 tools/gemm_lab 2>&1 | tail -20) | tee gpurun_out/${R}_gemm_lab.log
  (timeout 200 tools/gemm_lab clock 33 2>&1 | tail -10) | tee gpurun_out/${R}_gemm_clock.log
fi
if has soak; then  # production-shape soak of the default sampler (C3 window, 2000 launches, bitwise) + the stress-window soak of both versions
  (timeout 900 env WINDOW=c3 LAUNCHES=${SOAK_LAUNCHES:-2000} python tools/soak_corr.py 2>&1 | tail -6) | tee gpurun_out/${R}_soak_corr_c3.log
  (timeout 900 env REPS=24 QUIET=1 python tools/soak_corr.py 2>&1 | tail -6) | tee gpurun_out/${R}_soak_corr_stress.log
fi
if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q --durations=8 --tb=short > gpurun_out/${R}_pytest_gpu_full.log 2>&1
  grep -E "^(FAILED|ERROR|E  )|passed|failed" gpurun_out/${R}_pytest_gpu_full.log | head -60
  (grep -v "^E20\|^W20" gpurun_out/${R}_pytest_gpu_full.log | tail -25) > gpurun_out/${R}_pytest_gpu.log
fi
if has retest; then  # the tests named in $CTK_RETEST (a -k expression), full tracebacks
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -k "$CTK_RETEST" > gpurun_out/${R}_pytest_retest.log 2>&1
  grep -E "^(FAILED|ERROR|E  )|passed|failed" gpurun_out/${R}_pytest_retest.log | head -80
fi
if has bench; then
  (timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/${R}_bench_c3.err | tail -1) > gpurun_out/${R}_bench_c3.json   # the driver's command
  summ c3 gpurun_out/${R}_bench_c3.json
fi
if has more; then
  for w in c2_offline c4_online c5_shard c3_offline c3_offline_g40 c1_standin v2_sliding; do
    st=3; wu=1; [[ $w == c4_online ]] && { st=12; wu=3; }; [[ $w == c5_shard || $w == c3_offline* || $w == v2_sliding ]] && st=2
    (timeout 600 python bench.py --workload $w --steps $st --warmup $wu --no-cpu-baseline 2>gpurun_out/${R}_bench_$w.err | tail -1) > gpurun_out/${R}_bench_$w.json
    summ $w gpurun_out/${R}_bench_$w.json
  done
  (timeout 900 python bench.py --gpus 2 --single-device --dist-backend gloo --workload c5_shard --steps 1 --warmup 1 --no-cpu-baseline --no-profile 2>gpurun_out/${R}_bench_2rank.err | tail -1) > gpurun_out/${R}_bench_2rank_gloo_single_device_c5.json
  summ 2rank_c5 gpurun_out/${R}_bench_2rank_gloo_single_device_c5.json
fi
if has prof; then
  cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra-lines > /tmp/prof_bench.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/summarize_rocprof.py /tmp/prof gpurun_out/${R}_rocprof_kernel_stats.txt 2>&1 | tail -3
  sed -i "1i kernel sources sha256 $SRC_HASH (tools/gpu_session.sh prof: rocprofv3 --kernel-trace --stats over bench.py --steps 2 --warmup 1)" gpurun_out/${R}_rocprof_kernel_stats.txt
  tail -5 /tmp/prof_bench.log | cut -c1-300
  head -34 gpurun_out/${R}_rocprof_kernel_stats.txt
fi
if has sq; then
  cd /tmp && rm -rf /tmp/sq /tmp/ldsc
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-lines > /tmp/sq.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/summarize_counters.py /tmp/sq gpurun_out/${R}_sq_counters.txt | head -24
  sed -i "1i kernel sources sha256 $SRC_HASH" gpurun_out/${R}_sq_counters.txt
fi
