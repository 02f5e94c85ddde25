#!/bin/bash
# round 4: the refine_384 (configs[2]) and pseudo_masks (configs[4]) legs of bench.py, plain and under rocprofv3 --kernel-trace --stats
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
TAG=${1:-r4}
mkdir -p gpurun_out/$TAG
for leg in refine_384 pseudo_masks; do
  timeout -k 10 300 python3 bench.py --only-leg $leg > gpurun_out/$TAG/leg_$leg.json 2> gpurun_out/$TAG/leg_$leg.err || { echo "leg $leg failed"; tail -20 gpurun_out/$TAG/leg_$leg.err; exit 1; }
  tail -c 3000 gpurun_out/$TAG/leg_$leg.json; echo
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof_$leg -- python3 bench.py --only-leg $leg --no-cpu-baseline > gpurun_out/$TAG/prof_$leg.log 2>&1 || { echo "prof $leg failed"; tail -20 gpurun_out/$TAG/prof_$leg.log; exit 1; }
  F=$(find gpurun_out/$TAG/prof_$leg -name "*kernel_stats.csv" | head -1)
  cp "$F" gpurun_out/$TAG/kernel_stats_$leg.csv
  head -16 "$F"
  find gpurun_out/$TAG/prof_$leg -type f ! -name "*kernel_stats.csv" -delete
done
