#!/bin/bash
# Counter passes over the sampler micro-benchmark (tools/bench_corr.py): tools/pmc_corr.sh -> gpurun_out/pmc_corr.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf /tmp/pc$i
  ONLY=v1 REPS=3 timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pc$i -- python $R/tools/bench_corr.py > /tmp/pc$i.log 2>&1 || tail -5 /tmp/pc$i.log
  f=$(find /tmp/pc$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "corr_volume" not in r["Kernel_Name"]: continue
    k = "corr_volume_sh2" if "sh2" in r["Kernel_Name"] else "corr_volume_sh"
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    for c, v in sorted(d.items()):
        print(f"{k:36s} {c:28s} {v / n[(k, c)]:16.4g} per launch ({n[(k, c)]} launches)")
PY
done
