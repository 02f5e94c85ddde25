#!/bin/bash
# usage: bash scripts/pmc_run.sh <tag> "<counters>" <python script> ; counters in their own pass (no tracing domains)
set -e
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out
rocprofv3 --pmc $CNT --output-format csv -d gpurun_out/$TAG -- python3 "$@" > gpurun_out/$TAG.log 2>&1
echo rc=$?
F=$(ls -t $(find gpurun_out/$TAG -name "*counter_collection.csv") | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
