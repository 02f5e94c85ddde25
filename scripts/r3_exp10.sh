#!/bin/bash
# band walk of the evaluator kernels: parity, then timing against the raster walk (tuning library, SM_EVAL_BAND_MIN=0)
set -e
mkdir -p gpurun_out/r3j
python -m pytest tests/test_hip_eval.py tests/test_hip_evaluator.py -x -q 2>&1 | tail -5
T=$PWD/salient-object-detection_amd/lib/libselfmask_hip_tuning.so
echo "== product (band walk from H >= 2 mh)" | tee gpurun_out/r3j/eval_bench.log
python scripts/eval_bench.py 2>&1 | tee -a gpurun_out/r3j/eval_bench.log
echo "== tuning, raster walk for every image" | tee -a gpurun_out/r3j/eval_bench.log
SM_HIP_LIB=$T SM_EVAL_BAND_MIN=0 python scripts/eval_bench.py 2>&1 | tee -a gpurun_out/r3j/eval_bench.log
for u in 1 2; do echo "== tuning, band walk, $u unit(s) per wave" | tee -a gpurun_out/r3j/eval_bench.log; SM_HIP_LIB=$T SM_EVAL_UPW=$u python scripts/eval_bench.py 28 56 2>&1 | tee -a gpurun_out/r3j/eval_bench.log; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3j/prof -o ev -- python3 $GRAFT_REPO_ROOT/scripts/eval_bench.py 28 56 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3j/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
python bench.py --steps 50 --warmup 10 --no-other-shapes 2>/dev/null | tail -1 > gpurun_out/r3j/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3j/bench.json"))
print("bench", d["value"], d["ms_per_step"], "sustained", d.get("sustained", {}).get("value"), "e2e", d.get("end_to_end", {}).get("value"))
PY
python bench.py --quick --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quick full        ', d['value'], d['ms_per_step'])"
python bench.py --quick --forward-only --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quick forward-only', d['value'], d['ms_per_step'])"
