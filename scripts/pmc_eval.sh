#!/bin/bash
# SQ counters of the evaluator kernels on the bench workload's shape (scripts/eval_bench.py 28), counters in their own passes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"; mkdir -p gpurun_out
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
      "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
      "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS")
i=0
for S in "${SETS[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $S --output-format csv -d gpurun_out/sq_eval_$i -- python3 scripts/eval_bench.py 28 > gpurun_out/sq_eval_$i.log 2>&1 || echo "set $i failed (see gpurun_out/sq_eval_$i.log)"
done
python3 - <<'PY' > gpurun_out/r03_pmc_eval_counters.txt
import csv, glob, collections
print("SQ counters per launch of the evaluator kernels, B = 64, 28 x 28 masks, nq = 20, GT 300-400 px (rocprofv3 --pmc, one set per pass)\n")
per = collections.defaultdict(collections.OrderedDict)
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/sq_eval_{i}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "eval_" in k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            per[k][c] = sum(v) / len(v)
for k, vals in per.items():
    print("==", k)
    for c, v in vals.items():
        print(f"   {c:28s} {v:16.0f}")
PY
cat gpurun_out/r03_pmc_eval_counters.txt
