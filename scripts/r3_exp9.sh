#!/bin/bash
# round-3 experiment 9: wave priorities in the fused kernel's attention phase (tuning build, SM_QKV_PRIO = 0 | 1 | 2)
O=gpurun_out/r3o; mkdir -p $O
T=$PWD/salient-object-detection_amd/lib/libselfmask_hip_tuning.so
for p in 0 1 2; do echo "== SM_QKV_PRIO=$p"; SM_QKV_PRIO=$p python scripts/qkv_attn_bench.py 2>&1 | grep "fused "; SM_QKV_PRIO=$p python scripts/qkv_stamps.py 2>&1 | grep -v amdgpu.ids | grep "wave 0\|wave 4\|wave 3\|workgroup life"; done | tee $O/prio.log
one() { SM_HIP_LIB=$T python bench.py --quick --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do for p in 0 1 2; do echo -n "prio=$p  "; SM_QKV_PRIO=$p one; done; done | tee -a $O/prio.log
