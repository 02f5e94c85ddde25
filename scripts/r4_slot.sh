#!/bin/bash
# round 4: the new LDS slot image + swizzled epilogue staging: parity tests of everything that reads the image, the full-size shapes,
# then SQ counters (LDS bank conflicts) of the three shipped kernels and the bench A/B against the round-3 image
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
mkdir -p gpurun_out/r4
timeout -k 10 1000 python -m pytest tests/test_hip_gemm_w16.py tests/test_hip_qkv_attention.py tests/test_hip_forward.py tests/test_hip_bilateral.py -x -q 2>&1 | tee gpurun_out/r4/slot_tests.log | tail -8
grep -q "HSA_STATUS_ERROR" gpurun_out/r4/slot_tests.log && exit 1
tail -3 gpurun_out/r4/slot_tests.log | grep -q " passed" || exit 1
grep -q " failed" gpurun_out/r4/slot_tests.log && exit 1
cp gpurun_out/parity_ledger.json gpurun_out/r4/parity_ledger.json 2>/dev/null
for rep in 1 2; do
  for lib in "" "salient-object-detection_amd/lib/libselfmask_hip_slot_r3.so"; do
    SM_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --quick --steps 50 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('${lib:-new}', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], {k:v['avg_launch_us'] for k,v in d['roofline_other_kernels'].items() if 'gemm_w16m16_kernel<256' in k or 'qkv' in k})" | tee -a gpurun_out/r4/slot_ab.log
  done
done
