#!/bin/bash
# Everything DESIGN.md section 5 (round 4) quotes, from ONE box.  Outputs under gpurun_out/final4/ (copied into profiles/ as r04_*).
# usage: bash scripts/final_profiles_r04.sh [part]   part: a = tests + bench + kernel traces, b = PMC passes + sweeps (default: both)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"; mkdir -p gpurun_out/final4; O=gpurun_out/final4
PART=${1:-ab}
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
TAG=r04 bash scripts/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; cp gpurun_out/r04_pmc_traffic.json $O/
TAG=r04 bash scripts/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/r04_pmc_sq_counters.txt $O/ 2>/dev/null
rm -rf gpurun_out/sq_* gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
for s in 1 2 3 4; do python bench.py --quick --steps 60 --warmup 12 --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$s', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; done > $O/r04_streams.txt
python bench.py --quick --steps 60 --warmup 12 --forward-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forward-only (no evaluator kernels), 3 streams', d['value'], 'images/s')" >> $O/r04_streams.txt
python bench.py --quick --steps 60 --warmup 12 --zero-data 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('all-zero weights and images (diagnostic), 3 streams', d['value'], 'images/s')" >> $O/r04_streams.txt
cat $O/r04_streams.txt; tail -8 $O/pmc_traffic.log
fi
tail -c 600 $O/r04_bench_default.json 2>/dev/null; echo; head -6 $O/r04_kernel_stats.csv 2>/dev/null | cut -c1-150
