#!/bin/bash
# round-3 experiment 5: fused vs unfused attention by batch size (tuning build: SM_FUSED_QKV knob), forward only, 1 and 3 streams
O=gpurun_out/r3h; mkdir -p $O
T=$PWD/salient-object-detection_amd/lib/libselfmask_hip_tuning.so
one() { python bench.py --quick --forward-only "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for B in 1 4 16 24 32 48; do
  for S in 1 3; do
    echo -n "B=$B streams=$S fused   "; SM_HIP_LIB=$T one --batch $B --streams $S --steps 100 --warmup 20
    echo -n "B=$B streams=$S unfused "; SM_HIP_LIB=$T SM_FUSED_QKV=0 one --batch $B --streams $S --steps 100 --warmup 20
  done
done | tee $O/fused_vs_unfused_by_batch.log
python -m pytest tests/test_hip_gemm_w16.py tests/test_hip_qkv_attention.py -x -q -m gpu 2>&1 | tail -2
