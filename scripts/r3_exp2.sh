#!/bin/bash
# round-3 experiment 2: (a) attention step of the fused kernel as one basic block, (b) loader-half ring feed in the W16 GEMMs
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests/test_hip_qkv_attention.py tests/test_hip_gemm_w16.py tests/test_hip_forward.py tests/test_hip_evaluator.py tests/test_hip_inference.py tests/test_hip_voting.py tests/test_hip_eval.py -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
for r in m16x2L4 m16x2; do SM_QKV_RING=$r python scripts/qkv_attn_bench.py 2>&1 | grep "fused" ; done > $O/qkv_alone.log
SM_QKV_RING=m16x2L4 python scripts/qkv_stamps.py 2>&1 | grep -v amdgpu.ids > $O/qkv_stamps.log
python scripts/gemm_stamps.py 2>&1 | grep -v amdgpu.ids > $O/gemm_stamps.log
one() { python bench.py --quick --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
for i in 1 2 3; do
  echo -n "new       "; one
  echo -n "noloadh   "; SM_HIP_LIB=$PWD/salient-object-detection_amd/lib/libselfmask_hip_noloadh.so one
  echo -n "prev(r3b) "; SM_HIP_LIB=$PWD/salient-object-detection_amd/lib/libselfmask_hip_prev.so one
done > $O/pipeline_ab.log
cat $O/qkv_alone.log $O/qkv_stamps.log $O/gemm_stamps.log $O/pipeline_ab.log
