#!/bin/bash
# Everything DESIGN.md section 5 (round 4) quotes, from ONE box.  Outputs under gpurun_out/final4/ (copied into profiles/ as r04_*).
# usage: bash scripts/final_profiles_r04.sh [part]   part: a = tests + bench + kernel traces, b = PMC passes + sweeps, c = pseudo-mask path (default: ab)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"; mkdir -p gpurun_out/final4; O=gpurun_out/final4
PART=${1:-ab}
if [[ $PART == *b* ]]; then  # first: bench.py quotes profiles/r04_pmc_traffic.json only when it was taken on exactly these kernel sources
TAG=r04 bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; cp gpurun_out/r04_pmc_traffic.json $O/; cp gpurun_out/r04_pmc_traffic.json profiles/
fi
if [[ $PART == *a* ]]; then
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.log; cat $O/gpu_tests.log
cp gpurun_out/parity_ledger.json $O/r04_parity.json 2>/dev/null
python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
bash scripts/profile_bench.sh r04_streams1 --steps 20 --warmup 6 --streams 1 --no-graph > $O/profile.log 2>&1
cp $(find gpurun_out/r04_streams1 -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats.csv
grep '^{' gpurun_out/r04_streams1.log | tail -1 > $O/r04_streams1_bench.json
python scripts/trace_summary.py gpurun_out/r04_streams1 > $O/r04_forward_breakdown.txt 2>&1
bash scripts/profile_bench.sh r04_p8 --steps 10 --warmup 4 --streams 1 --no-graph --patch 8 --batch 16 > $O/profile_p8.log 2>&1
cp $(find gpurun_out/r04_p8 -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_vit_s8_224.csv
bash scripts/profile_bench.sh r04_384 --steps 10 --warmup 4 --streams 1 --no-graph --size 384 --batch 32 > $O/profile_384.log 2>&1
cp $(find gpurun_out/r04_384 -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_vit_s16_384.csv
bash scripts/profile_bench.sh r04_b1 --steps 20 --warmup 6 --streams 1 --no-graph --batch 1 > $O/profile_b1.log 2>&1
python scripts/trace_summary.py gpurun_out/r04_b1 > $O/r04_forward_breakdown_b1.txt 2>&1
for leg in refine_384 pseudo_masks; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_$leg -- python3 bench.py --only-leg $leg --no-cpu-baseline > $O/prof_$leg.log 2>&1
  cp $(find gpurun_out/r04_$leg -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats_$leg.csv
done
rm -rf gpurun_out/r04_streams1 gpurun_out/r04_p8 gpurun_out/r04_384 gpurun_out/r04_b1 gpurun_out/r04_refine_384 gpurun_out/r04_pseudo_masks
fi
if [[ $PART == *b* ]]; then
TAG=r04 bash scripts/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/r04_pmc_sq_counters.txt $O/ 2>/dev/null
rm -rf gpurun_out/sq_* gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
for s in 1 2 3 4; do python bench.py --quick --steps 60 --warmup 12 --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$s', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; done > $O/r04_streams.txt
python bench.py --quick --steps 60 --warmup 12 --forward-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward-only (no evaluator kernels), 3 streams', d['value'], 'images/s')" >> $O/r04_streams.txt
python bench.py --quick --steps 60 --warmup 12 --zero-data 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all-zero weights and images (diagnostic), 3 streams', d['value'], 'images/s')" >> $O/r04_streams.txt
cat $O/r04_streams.txt; tail -8 $O/pmc_traffic.log
fi
if [[ $PART == *c* ]]; then   # the pseudo-mask path: batch x streams sweep, the clusterer by size / filter degree / phase, its SQ counters
bash scripts/r4_pseudo_batches.sh > /dev/null 2>&1; cp gpurun_out/r4/pseudo_batches.log $O/r04_pseudo_masks_by_batch.log
bash scripts/r4_spectral_sizes.sh > /dev/null 2>&1; cp gpurun_out/r4/spectral_sizes.log $O/r04_spectral_by_points.log
python scripts/spectral_degree.py 28 16 > $O/r04_spectral_degree_scenes.log 2>&1
python scripts/spectral_degree.py bench 128 > $O/r04_spectral_degree_bench.log 2>&1
for g in 28 32 44; do SM_HIP_LIB=$ROOT/salient-object-detection_amd/lib/libselfmask_hip_spstamps.so python scripts/spectral_stamps.py $g; done > $O/r04_spectral_phases.log 2>&1
SM_HIP_LIB=$ROOT/salient-object-detection_amd/lib/libselfmask_hip_spstamps.so python scripts/spectral_bench_features.py 4 >> $O/r04_spectral_phases.log 2>&1
bash scripts/r4_spectral_pmc.sh > /dev/null 2>&1; cp gpurun_out/r4/spectral_sq_counters.txt $O/r04_spectral_sq_counters.txt
tail -3 $O/r04_pseudo_masks_by_batch.log | cut -c1-200; cat $O/r04_spectral_by_points.log
fi
tail -c 600 $O/r04_bench_default.json 2>/dev/null; echo; head -6 $O/r04_kernel_stats.csv 2>/dev/null | cut -c1-150
