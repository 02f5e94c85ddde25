#!/bin/bash
# Everything DESIGN.md section 5 (round 3) quotes, from ONE box.  Outputs under gpurun_out/final3/ (copied into profiles/ as r03_*).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/final3; O=gpurun_out/final3
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.log; cat $O/gpu_tests.log
python bench.py > $O/r03_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
# kernel trace of the single-stream eager run (taps vs rocprofv3 agreement), headline shape and the two other shapes
bash scripts/profile_bench.sh r03_streams1 --steps 20 --warmup 6 --streams 1 --no-graph > $O/profile.log 2>&1
cp $(find gpurun_out/r03_streams1 -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats.csv
grep '^{' gpurun_out/r03_streams1.log | tail -1 > $O/r03_streams1_bench.json
python scripts/trace_summary.py gpurun_out/r03_streams1 > $O/r03_forward_breakdown.txt 2>&1
bash scripts/profile_bench.sh r03_p8 --steps 10 --warmup 4 --streams 1 --no-graph --patch 8 --batch 16 > $O/profile_p8.log 2>&1
cp $(find gpurun_out/r03_p8 -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats_vit_s8_224.csv
python scripts/trace_summary.py gpurun_out/r03_p8 > $O/r03_forward_breakdown_vit_s8_224.txt 2>&1
bash scripts/profile_bench.sh r03_384 --steps 10 --warmup 4 --streams 1 --no-graph --size 384 --batch 32 > $O/profile_384.log 2>&1
cp $(find gpurun_out/r03_384 -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats_vit_s16_384.csv
python scripts/trace_summary.py gpurun_out/r03_384 > $O/r03_forward_breakdown_vit_s16_384.txt 2>&1
# HBM traffic per launch (separate FETCH_SIZE / WRITE_SIZE passes), stamped with the kernel-source hash
bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; cp gpurun_out/r03_pmc_traffic.json $O/
# stream sweep, fused kernel alone + stamps, GEMM stamps
for s in 1 2 3 4; do python bench.py --quick --steps 60 --warmup 12 --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$s', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; done > $O/r03_streams.txt
python bench.py --quick --steps 60 --warmup 12 --forward-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward-only (no evaluator kernels), 3 streams', d['value'], 'images/s')" >> $O/r03_streams.txt
python bench.py --quick --steps 60 --warmup 12 --zero-data 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all-zero weights and images (diagnostic), 3 streams', d['value'], 'images/s')" >> $O/r03_streams.txt
for r in m16x2L4 m16x2; do SM_QKV_RING=$r python scripts/qkv_attn_bench.py 2>&1 | grep SM_QKV; done > $O/r03_qkv_attention_bench.log
for r in m16x2L4 m16x2; do SM_QKV_RING=$r python scripts/qkv_stamps.py 2>&1 | grep -v amdgpu.ids; done > $O/r03_qkv_stamps.txt
python scripts/gemm_stamps.py 2>&1 | grep -v amdgpu.ids > $O/r03_gemm_stamps.txt
bash scripts/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/r03_pmc_sq_counters.txt $O/ 2>/dev/null
# evaluator kernels alone: band walk (product) against the raster walk (tuning library), per-kernel times, SQ counters
T=$ROOT/salient-object-detection_amd/lib/libselfmask_hip_tuning.so
{ echo "== product (band walk from H >= 2 mh)"; python scripts/eval_bench.py 2>&1 | grep -v amdgpu.ids
  echo "== tuning library, raster walk for every image (SM_EVAL_BAND_MIN=0)"; SM_HIP_LIB=$T SM_EVAL_BAND_MIN=0 python scripts/eval_bench.py 2>&1 | grep -v amdgpu.ids
  for u in 1 2; do echo "== tuning library, band walk, $u unit(s) per wave (product: 4)"; SM_HIP_LIB=$T SM_EVAL_UPW=$u python scripts/eval_bench.py 28 56 2>&1 | grep -v amdgpu.ids; done; } > $O/r03_eval_band_walk_ab.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/final3/eval_prof -o ev -- python3 $ROOT/scripts/eval_bench.py 28 > /dev/null 2>&1 )
cp $(find gpurun_out/final3/eval_prof -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats_evaluator_alone.csv 2>/dev/null
bash scripts/pmc_eval.sh > $O/pmc_eval.log 2>&1; cp gpurun_out/r03_pmc_eval_counters.txt $O/ 2>/dev/null
tail -c 400 $O/r03_bench_default.json; echo; cat $O/r03_streams.txt; head -6 $O/r03_kernel_stats.csv | cut -c1-150; head -8 $O/r03_forward_breakdown.txt; tail -8 $O/pmc_traffic.log
