#!/bin/bash
# SQ counters of the shipped kernels, alone (guide's recipe: counters in their own passes, no tracing domains): wave cycles,
# waits, MFMA-busy cycles, LDS conflicts, clock.  Three kernels x three counter sets -> gpurun_out/${TAG:-r04}_pmc_sq_counters.txt
#   fc1  = gemm_w16m16<256,256>, M 12608, N 1536, K 384, GELU + F16X2 out   (variant 40)
#   proj = gemm_w16m16<256,128,3,4,4,4>, N 384, K 384, residual             (variant 47)
#   fused QKV + attention (B 64, N 197)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"; mkdir -p gpurun_out
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"
      "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
      "GRBM_GUI_ACTIVE GRBM_COUNT SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
i=0
for S in "${SETS[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $S --output-format csv -d gpurun_out/sq_fc1_$i -- python3 scripts/one_gemm_w16.py 12608 1536 384 40 gelu 1 > gpurun_out/sq_fc1_$i.log 2>&1
  rocprofv3 --pmc $S --output-format csv -d gpurun_out/sq_proj_$i -- python3 scripts/one_gemm_w16.py 12608 384 384 47 residual 0 > gpurun_out/sq_proj_$i.log 2>&1
  rocprofv3 --pmc $S --output-format csv -d gpurun_out/sq_qkv_$i -- python3 scripts/qkv_attn_bench.py > gpurun_out/sq_qkv_$i.log 2>&1
done
python3 - <<'PY' > gpurun_out/${TAG:-r04}_pmc_sq_counters.txt
import csv, glob, collections
want = {"fc1": "gemm_w16m16_kernel<256, 256", "proj": "gemm_w16m16_kernel<256, 128", "qkv": "qkv_attention_m16_kernel"}
print("SQ counters per launch (average over the launches of one process; rocprofv3 --pmc, one counter set per pass;")
print("SQ_* cycle counters are summed over waves / SIMDs as the hardware reports them, GRBM_GUI_ACTIVE over the 8 XCDs)\n")
for tag, frag in want.items():
    vals = collections.OrderedDict()
    n = 0
    for i in (1, 2, 3):
        for f in glob.glob(f"gpurun_out/sq_{tag}_{i}/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if frag in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    vgpr, lds, wg, grid = r["VGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"]
            for c, v in agg.items():
                vals[c] = sum(v) / len(v)
                n = len(v)
    if not vals:
        print(tag, "no data"); continue
    print(f"== {tag}: {frag}...  grid {grid} threads, workgroup {wg}, VGPRs {vgpr}, LDS {lds} B, {n} launches averaged")
    for c, v in vals.items():
        print(f"   {c:32s} {v:16.0f}")
    wc = vals.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in vals: print(f"   {c + ' / SQ_WAVE_CYCLES':48s} {vals[c] / wc:6.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        kc = vals["GRBM_GUI_ACTIVE"] / 8.0  # kernel duration in shader-clock cycles (the counter sums the 8 XCDs)
        print(f"   {'kernel duration (GRBM_GUI_ACTIVE / 8 XCDs), cycles':48s} {kc:9.0f}")
        print(f"   {'matrix-pipe utilisation = MFMA_BUSY / (1024 SIMDs x duration)':48s} {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * kc):6.3f}")
    if "SQ_LDS_BANK_CONFLICT" in vals and vals.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   {'SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE':48s} {vals['SQ_LDS_BANK_CONFLICT'] / vals['SQ_LDS_IDX_ACTIVE']:6.3f}")
    if "SQ_INSTS_VALU" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
        mf = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 16.0  # v_mfma_f32_16x16x32_f16: 16 cycles each
        print(f"   {'MFMA instructions (MFMA_BUSY / 16)':48s} {mf:9.0f}")
        print(f"   {'VALU instructions (MFMA included) per MFMA':48s} {vals['SQ_INSTS_VALU'] / mf:6.2f}")
    print()
PY
cat gpurun_out/${TAG:-r04}_pmc_sq_counters.txt
