#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/r4/all_gpu_tests.log | tail -12
cp gpurun_out/parity_ledger.json gpurun_out/r4/parity_ledger.json 2>/dev/null
grep -q "HSA_STATUS_ERROR" gpurun_out/r4/all_gpu_tests.log && exit 1
tail -3 gpurun_out/r4/all_gpu_tests.log | grep -q " passed" || exit 1
python3 bench.py --quick --steps 50 --warmup 10 --batch 1 --streams 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('B=1 forward+eval ms/step', d['ms_per_step'])"
python3 - <<'PY'
import sys, os, json
sys.path[:0] = ['salient-object-detection_amd', '.']
import torch, bench
torch.cuda.set_device(0)
w = bench.Workload(torch.device('cuda', 0), 16, 224, 1, streams=1)
print(json.dumps({"serving": bench.serving_latency(w.model, torch.device('cuda', 0))}))
PY
