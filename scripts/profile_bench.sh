#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench run; summaries land under gpurun_out/<tag>/ (copy the
# *_kernel_stats.csv you want judged into profiles/).   usage: bash scripts/profile_bench.sh <tag> [bench args]
set -e
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 bench.py --quick "$@" > gpurun_out/$TAG.log 2>&1
echo "rc=$?"
tail -1 gpurun_out/$TAG.log
F=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
echo "stats: $F"
head -25 "$F"
