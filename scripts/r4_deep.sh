#!/bin/bash
# round 4: A/B harness for the two small-launch experiments (ring-of-six 64 x 64 GEMM, then attention with all key chunks staged at once): the product library against an experiment build that disables the change; both changes lost and live in the tuning build now (profiles/r04_deep_ring_ab.log, r04_attn_all_ab.log)
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_ops.py tests/test_hip_forward.py tests/test_hip_inference.py -x -q 2>&1 | tee gpurun_out/r4/attn_all_tests.log | tail -5
tail -3 gpurun_out/r4/attn_all_tests.log | grep -q " passed" || exit 1
grep -q " failed" gpurun_out/r4/attn_all_tests.log && exit 1
for rep in 1 2; do
for lib in "" "salient-object-detection_amd/lib/libselfmask_hip_noall.so"; do
SM_HIP_LIB=$lib python3 - <<PY
import sys, os, json, time
sys.path[:0] = ['salient-object-detection_amd', '.']
import torch, bench
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
tag = "attention: ring of two          " if os.environ.get("SM_HIP_LIB") else "attention: all chunks at once   "
out = {}
for B in (1, 2, 8, 64):
    w = bench.Workload(dev, 16, 224, B, streams=1 if B < 64 else 3)
    w.prime(6)
    dt = w.timed(60 if B < 64 else 30)
    out[B] = round(dt / (60 if B < 64 else 30) * 1e3, 4)
    if B == 1:
        s = bench.serving_latency(w.model, dev)
        out["serving_p50"] = s["p50_ms"]
    del w
print(tag, "ms per step (forward + evaluator) at batch 1 / 2 / 8 (one stream), 64 (three streams):", out)
PY
done; done | tee gpurun_out/r4/attn_all_ab.log
