#!/bin/bash
# round-3 experiment 3: attention step as one basic block with the compiler-visible split; LayerNorm ablation (timing only)
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests/test_hip_qkv_attention.py tests/test_hip_ops.py -x -q -m gpu > $O/tests_qkv.log 2>&1; echo "qkv pytest rc=$?"; tail -3 $O/tests_qkv.log
for r in m16x2L4 m16x2; do SM_QKV_RING=$r python scripts/qkv_attn_bench.py 2>&1 | grep "fused" ; done > $O/qkv_alone.log
python scripts/qkv_stamps.py 2>&1 | grep -v amdgpu.ids > $O/qkv_stamps.log
one() { python bench.py --quick --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
T=$PWD/salient-object-detection_amd/lib/libselfmask_hip_tuning.so
for i in 1 2 3; do
  echo -n "new            "; one
  echo -n "tuning         "; SM_HIP_LIB=$T one
  echo -n "tuning, no LN  "; SM_HIP_LIB=$T SM_ABLATE_LN=1 one
done > $O/pipeline_ab.log
cat $O/qkv_alone.log $O/qkv_stamps.log $O/pipeline_ab.log
python -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; echo "all pytest rc=$?"; tail -3 $O/tests_all.log
