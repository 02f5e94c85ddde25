#!/bin/bash
# round 4: 256 x 192 tiles (400 instead of 300 for fc1: 1.56 rounds of 0.75-size tiles instead of 1.17 rounds in 2) against 256 x 256
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_gemm_w16.py -x -q 2>&1 | tee gpurun_out/r4/t192_tests.log | tail -4
tail -3 gpurun_out/r4/t192_tests.log | grep -q " passed" || exit 1
for rep in 1 2 3; do
  for lib in "" "salient-object-detection_amd/lib/libselfmask_hip_t192.so"; do
    for s in 3 1; do
    SM_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --quick --steps 50 --warmup 10 --streams $s 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('${lib:-256x256 (product)}', 'streams $s', d['value'], 'images/s;', d['roofline']['kernel'], d['roofline']['avg_launch_us'], 'us', d['roofline']['frac'])"
    done
  done
done | tee gpurun_out/r4/t192_ab.log
