#!/bin/bash
# In-pipeline A/B of two builds of the library in ONE gpurun call (box-to-box variance is 2-5 %): alternates
# bench.py --quick between lib/libselfmask_hip.so and the library given as $1 (SM_HIP_LIB override), $2 rounds.
# The comparison library: `cp lib/libselfmask_hip.so lib/libselfmask_hip_prev.so` before rebuilding, or `git archive <rev>
# salient-object-detection_amd/csrc include | tar -x -C /tmp/prev` + hipcc -c each source + one -shared link.
other=${1:-salient-object-detection_amd/lib/libselfmask_hip_prev.so}
rounds=${2:-2}
one() { python bench.py --quick --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in $(seq $rounds); do
  echo -n "new  "; one
  echo -n "prev "; SM_HIP_LIB=$PWD/$other one
done
