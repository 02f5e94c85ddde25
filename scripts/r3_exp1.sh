#!/bin/bash
# round-3 experiment 1: loader-half fused kernel (SM_QKV_RING=m16x2L4 / m16x3L4) vs all-wave loading, alone + stamps + pipeline;
# spread DMA issue in the GEMM K loop (libselfmask_hip_spread.so) in the pipeline
O=gpurun_out/r3c; mkdir -p $O
for r in m16x2 m16x2L4 m16x3 m16x3L4; do SM_QKV_RING=$r python scripts/qkv_attn_bench.py 2>&1 | grep "fused" ; done > $O/qkv_alone.log
for r in m16x2 m16x2L4 m16x3L4; do SM_QKV_RING=$r python scripts/qkv_stamps.py 2>&1 | grep -v amdgpu.ids; done > $O/qkv_stamps.log
one() { python bench.py --quick --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  echo -n "base      "; one
  echo -n "L4        "; SM_QKV_RING=m16x2L4 one
  echo -n "L4x3      "; SM_QKV_RING=m16x3L4 one
  echo -n "spread    "; SM_HIP_LIB=$PWD/salient-object-detection_amd/lib/libselfmask_hip_spread.so one
done > $O/pipeline_ab.log
cat $O/qkv_alone.log $O/qkv_stamps.log $O/pipeline_ab.log
