#!/bin/bash
# One gpurun call: parity tests, GEMM micro-bench, headline bench, PMC counters.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
(CTK_GEMM_TILE=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm or forward_window" 2>&1 | tail -15) > gpurun_out/pytest_gpu_tile2.log
(MODES=f16x3,sh,sh2sh ROUNDS=5 timeout 300 python tools/bench_gemm.py 2>&1 | tail -70) > gpurun_out/bench_gemm_t1.log
(CTK_GEMM_TILE=2 MODES=sh ROUNDS=5 timeout 300 python tools/bench_gemm.py 2>&1 | tail -50) > gpurun_out/bench_gemm_t2.log
(timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench.err | tail -1) > gpurun_out/bench_t1.json
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && MODES=sh SHAPES=corr_fc1,q_all,fc1_all,fc2_all ROUNDS=2 timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/pmc$i -- python $GRAFT_REPO_ROOT/tools/bench_gemm.py > /tmp/pmc$i.log 2>&1)
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/summarize_pmc.py "$f" > gpurun_out/pmc_gemm_pass$i.txt 2>&1; else tail -5 /tmp/pmc$i.log > gpurun_out/pmc_gemm_pass$i.txt; fi
done
tail -4 gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu_tile2.log; cat gpurun_out/bench_gemm_t1.log | tail -40; cat gpurun_out/bench_gemm_t2.log;  head -c 400 gpurun_out/bench_t1.json; echo; cat gpurun_out/pmc_gemm_pass*.txt
