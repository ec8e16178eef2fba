#!/bin/bash
# Round-4 GPU session: tools/gpu_session.sh [tests] [bench] [more] [prof] [pmc] [sq] [lab]   (default: tests bench)
# One gpurun call = one box: everything wanted from it is listed here; outputs go to gpurun_out/r04_*.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r04
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
if has lab; then
  (timeout 600 tools/gemm_lab 2>&1 | tail -20) | tee gpurun_out/${R}_gemm_lab.log
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
if has ovab; then  # CTK_OVERLAP bit 4: time-block q projection beside the kv projection (aux stream), A/B/A
  for v in 0 4 0 4; do
    (CTK_OVERLAP=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extra-lines 2>gpurun_out/${R}_ovab_$v.err | tail -1) > gpurun_out/${R}_bench_c3_overlap_$v.json
    python -c "import json,sys; d=json.load(open('gpurun_out/${R}_bench_c3_overlap_$v.json')); print('CTK_OVERLAP=$v', d['value'], d['ms_per_step'], json.dumps(d['parity']['timed_step'])[-330:-200])"
  done
fi
if has latetrace; then  # kernel trace of one step with overlap mode 8: do the two queues overlap in time?
  cd /tmp && rm -rf /tmp/kt
  CTK_OVERLAP=8 CTK_SIDE_CUS=160 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-lines > /tmp/kt.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/trace_overlap.py /tmp/kt | tee gpurun_out/${R}_overlap8_trace.txt
fi
if has lateab; then  # CTK_OVERLAP bit 8: points<-virtual query projection beside the small launches of the virtual-track chain, on CTK_SIDE_CUS CUs
  for v in "0 192" "8 192" "8 224" "8 160" "0 192" "8 192" "8 128"; do
    set -- $v
    (CTK_OVERLAP=$1 CTK_SIDE_CUS=$2 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-extra-lines 2>gpurun_out/${R}_lateab.err | tail -1) > gpurun_out/${R}_bench_c3_late_$1_$2.json
    python -c "import json; d=json.load(open('gpurun_out/${R}_bench_c3_late_$1_$2.json')); print('CTK_OVERLAP=$1 CTK_SIDE_CUS=$2', d['value'], d['ms_per_step'], d['parity']['timed_step']['coords_px'], d['parity']['timed_step']['vis_logit'])"
  done
fi
if has deepab; then  # small-M GEMM kernel: round-3 ring (CTK_GEMM_DEEP64=4) vs 8 slots / 2 K-tiles per iteration (82, default); C4 also with CTK_OVERLAP=2
  for v in 4 82 4 82; do
    (CTK_GEMM_DEEP64=$v timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-lines 2>gpurun_out/${R}_deepab_c3_$v.err | tail -1) > gpurun_out/${R}_bench_c3_deep64_$v.json
    python - gpurun_out/${R}_bench_c3_deep64_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C3 CTK_GEMM_DEEP64=" + sys.argv[2], d["value"], d["ms_per_step"], "gemm_sh_64 rows:", [(k["name"], k["avg_us"]) for k in d["kernels"] if k["name"].startswith("gemm_sh_64")],
      "parity", d["parity"]["timed_step"]["coords_px"], d["parity"]["timed_step"]["vis_logit"])
PY
  done
  for v in "4 0" "82 0" "82 2" "4 0" "82 0" "82 2"; do
    set -- $v
    (CTK_GEMM_DEEP64=$1 CTK_OVERLAP=$2 timeout 600 python bench.py --workload c4_online --steps 24 --warmup 6 --no-cpu-baseline --no-profile 2>gpurun_out/${R}_deepab_c4.err | tail -1) > gpurun_out/${R}_bench_c4_deep64_$1_ov$2.json
    python -c "import json; d=json.load(open('gpurun_out/${R}_bench_c4_deep64_$1_ov$2.json')); print('C4 DEEP64=$1 OVERLAP=$2', d['value'], d['ms_per_step'])"
  done
fi
if has haloab; then  # 3x3 convolutions: halo kernel (default) vs conv_pp128_kernel (CTK_CONV_HALO=0), C2 and the encoder alone
  for v in 0 1 0 1; do
    (CTK_CONV_HALO=$v timeout 600 python bench.py --workload c2_offline --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/${R}_haloab_$v.err | tail -1) > gpurun_out/${R}_bench_c2_halo_$v.json
    python - gpurun_out/${R}_bench_c2_halo_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("CTK_CONV_HALO=" + sys.argv[2], d["ms_per_step"], "ms/step; parity", json.dumps(d.get("parity", {}).get("timed_step", {}))[-420:-300])
for k in d["kernels"]:
    if k["name"].startswith(("conv_", "enc_")): print("    ", k["name"], k["launches"], k["total_ms"], k["avg_us"], k["tflops"])
PY
  done
fi
if has convab; then  # encoder: second column phase skipped for the 64-channel layers (default) vs round-3 behaviour
  for v in 1 0; do
    (CTK_CONV_PH1=$v timeout 600 python bench.py --workload c2_offline --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/${R}_convab_$v.err | tail -1) > gpurun_out/${R}_bench_c2_convph1_$v.json
    python - gpurun_out/${R}_bench_c2_convph1_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("CTK_CONV_PH1=" + sys.argv[2], d["ms_per_step"], "ms/step; parity", json.dumps(d.get("parity", {}).get("timed_step", {}))[:260])
for k in d["kernels"]:
    if k["name"].startswith(("conv_pp128", "enc_")): print("    ", k["name"], k["launches"], k["total_ms"], k["avg_us"])
PY
  done
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
  tail -5 /tmp/prof_bench.log | cut -c1-300
  head -34 gpurun_out/${R}_rocprof_kernel_stats.txt
fi
if has corrpmc; then  # SQ counters of the sampler alone (tools/bench_corr.py micro-benchmark, three separate --pmc passes)
  bash tools/pmc_corr.sh > gpurun_out/${R}_pmc_corr.txt 2>&1
  cd $GRAFT_REPO_ROOT
  grep corr_volume_sh gpurun_out/${R}_pmc_corr.txt | head -30
fi
if has sqc2; then  # SQ counters per kernel on the C2 workload (encoder kernels visible)
  cd /tmp && rm -rf /tmp/sqc2 /tmp/sqc2b
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/sqc2 -- python $GRAFT_REPO_ROOT/bench.py --workload c2_offline --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /tmp/sqc2.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d /tmp/sqc2b -- python $GRAFT_REPO_ROOT/bench.py --workload c2_offline --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /tmp/sqc2b.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/summarize_counters.py /tmp/sqc2 gpurun_out/${R}_sq_counters_c2.txt | grep -E "kernel|conv|enc_" | head -20
  python tools/summarize_counters.py /tmp/sqc2b gpurun_out/${R}_lds_counters_c2.txt | grep -E "kernel|conv" | head -12
fi
if has sq; then
  cd /tmp && rm -rf /tmp/sq /tmp/ldsc
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-lines > /tmp/sq.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/summarize_counters.py /tmp/sq gpurun_out/${R}_sq_counters.txt | head -24
fi
