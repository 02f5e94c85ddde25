#!/bin/bash
# round 4: the multi-workgroup bilateral solver: parity tests, then the refine_384 leg plain and under rocprofv3
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
TAG=${1:-r4b}
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_hip_bilateral.py tests/test_hip_evaluator.py -x -q -s 2>&1 | tee gpurun_out/$TAG/bilateral_tests.log | tail -25
grep -q "HSA_STATUS_ERROR" gpurun_out/$TAG/bilateral_tests.log && exit 1
tail -3 gpurun_out/$TAG/bilateral_tests.log | grep -q "passed" || exit 1
grep -q "failed" gpurun_out/$TAG/bilateral_tests.log && exit 1
timeout -k 10 300 python3 bench.py --only-leg refine_384 > gpurun_out/$TAG/leg_refine_384.json 2> gpurun_out/$TAG/leg_refine_384.err || { tail -20 gpurun_out/$TAG/leg_refine_384.err; exit 1; }
cat gpurun_out/$TAG/leg_refine_384.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python3 bench.py --only-leg refine_384 --no-cpu-baseline > gpurun_out/$TAG/prof.log 2>&1 || { tail -20 gpurun_out/$TAG/prof.log; exit 1; }
F=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/$TAG/kernel_stats_refine_384.csv
head -40 "$F" | cut -c1-200
find gpurun_out/$TAG/prof -type f -delete
