#!/bin/bash
# Everything DESIGN.md section 5 quotes, from ONE box: default bench, kernel-trace stats of the single-stream eager run,
# PMC traffic, stream sweep, fused-kernel A/B.  Outputs under gpurun_out/final/ (copy into profiles/ as r02_*).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/final; O=gpurun_out/final
python bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
bash scripts/profile_bench.sh r02_streams1 --steps 20 --warmup 6 --streams 1 --no-graph > $O/profile.log 2>&1
cp $(find gpurun_out/r02_streams1 -name "*kernel_stats.csv" | head -1) $O/r02_kernel_stats.csv
grep '^{' gpurun_out/r02_streams1.log | tail -1 > $O/r02_streams1_bench.json
python scripts/trace_summary.py gpurun_out/r02_streams1 > $O/r02_forward_breakdown.txt 2>&1
bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; cp gpurun_out/r02_pmc_traffic.json $O/
for s in 1 2 3 4; do python bench.py --quick --steps 60 --warmup 12 --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$s', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; done > $O/r02_streams.txt
python bench.py --quick --steps 60 --warmup 12 --forward-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward-only (no evaluator kernels), 3 streams', d['value'], 'images/s')" >> $O/r02_streams.txt
for r in m16x2 m16x3 32x2 32x3 16x2 16x6; do SM_QKV_RING=$r python scripts/qkv_attn_bench.py 2>&1 | grep SM_QKV; done > $O/r02_qkv_attention_bench.log
# in-kernel stamps (tuning build): where a tile / a workgroup spends its life, alone and inside the 3-stream pipeline
python scripts/gemm_stamps.py 2>&1 | grep -v amdgpu.ids > $O/r02_gemm_stamps.txt
python scripts/qkv_stamps.py 2>&1 | grep -v amdgpu.ids > $O/r02_qkv_stamps.txt
(TILES=300 python scripts/pipeline_stamps.py 1536 384 2>/dev/null | grep -v "^{"; python scripts/pipeline_stamps.py 384 1536 2>/dev/null | grep -v "^{"
 python scripts/pipeline_stamps.py 384 384 2>/dev/null | grep -v "^{") > $O/r02_pipeline_stamps.txt
VARIANTS=40,41,42,45,46,47,48 python scripts/gemm_w16_sweep.py > $O/r02_gemm_w16_sweep_m16.log 2>&1
tail -c 600 $O/r02_bench_default.json; echo; cat $O/r02_streams.txt; head -8 $O/r02_kernel_stats.csv | cut -c1-120; head -12 $O/r02_forward_breakdown.txt; tail -8 $O/pmc_traffic.log
