#!/bin/bash
# SQ counters of the spectral eigen-solver alone (16 synthetic scenes, n = 784): where its cycles go (LDS conflicts, waits)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"; mkdir -p gpurun_out/r4
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
      "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
      "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC")
i=0
for S in "${SETS[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $S --output-format csv -d gpurun_out/r4/sqsp_$i -- python3 scripts/spectral_degree.py 28 16 > gpurun_out/r4/sqsp_$i.log 2>&1
done
python3 - <<'PY' > gpurun_out/r4/spectral_sq_counters.txt
import csv, glob, collections
vals = collections.OrderedDict()
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/r4/sqsp_{i}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "spectral_embed_kernel" in r["Kernel_Name"] and r["Grid_Size"] == str(16 * 512):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in agg.items():
            vals[c] = (sum(v) / len(v), len(v))
print("spectral_embed_kernel, 16 images of n = 784 (16 workgroups of 512): SQ counters per launch")
for c, (v, n) in vals.items():
    print(f"  {c:28s} {v:16.0f}   ({n} launches)")
PY
cat gpurun_out/r4/spectral_sq_counters.txt
rm -rf gpurun_out/r4/sqsp_*
