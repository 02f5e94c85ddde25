#!/bin/bash
# round-3 experiment 4: where batch-1 latency goes (per-kernel taps at B = 1), fused vs unfused attention at B = 1; host CPU facts
O=gpurun_out/r3g; mkdir -p $O
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms/step')
r=d['roofline']; ks={r['kernel']:r}; ks.update(d['roofline_other_kernels'])
tot=0
for k,v in ks.items():
    print('   %-50s x%5.1f  %7.2f us  = %7.1f us' % (k, v['launches_per_forward'], v['avg_launch_us'], v['launches_per_forward']*v['avg_launch_us'])); tot+=v['launches_per_forward']*v['avg_launch_us']
print('   tapped total', round(tot,1), 'us')
"; }
python bench.py --quick --batch 1 --streams 1 --steps 200 --warmup 20 --forward-only 2>/dev/null | show "B=1 fused   "
SM_FUSED_QKV=0 python bench.py --quick --batch 1 --streams 1 --steps 200 --warmup 20 --forward-only 2>/dev/null | show "B=1 unfused "
python bench.py --quick --batch 8 --streams 1 --steps 100 --warmup 20 --forward-only 2>/dev/null | show "B=8 fused   "
SM_FUSED_QKV=0 python bench.py --quick --batch 8 --streams 1 --steps 100 --warmup 20 --forward-only 2>/dev/null | show "B=8 unfused "
for w in 8 15 24 32; do SM_DECODE_WORKERS=$w python - <<PY
import sys, time, os, tempfile
sys.path[:0]=['salient-object-detection_amd','.']
from selfmask_amd import datasets as DS
from selfmask_amd.pipeline import PrefetchingLoader
root=tempfile.mkdtemp(); DS.write_synthetic_dataset(root,"duts",512,seed=7); ds=DS.get_dataset(root,"duts")
idx=list(range(512))*4
t0=time.perf_counter(); n=0; t1=None
for rgbs,gts,i in PrefetchingLoader(ds, idx, 64, depth=4):
    if t1 is None: t1=time.perf_counter(); n0=len(rgbs)
    n+=len(rgbs)
t2=time.perf_counter()
print("decode only, $w workers: %.0f img/s after the first batch"%((n-n0)/(t2-t1)))
PY
done
