#!/bin/bash
# round-3 experiment 6: attention_f16x2 with compiler-scheduled transposed reads: parity + the ViT-S/8 and 384^2 legs
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests/test_hip_ops.py tests/test_hip_forward.py -x -q -m gpu 2>&1 | tail -2
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms/step')
r=d['roofline']; ks={r['kernel']:r}; ks.update(d['roofline_other_kernels'])
for k,v in ks.items():
    print('   %-50s x%5.1f  %7.2f us  frac %s' % (k, v['launches_per_forward'], v['avg_launch_us'], v.get('frac')))
"; }
python bench.py --quick --steps 20 --warmup 5 --patch 8 --batch 16 2>/dev/null | show "P8 224 B16 "
python bench.py --quick --steps 20 --warmup 5 --size 384 --batch 32 2>/dev/null | show "P16 384 B32"
python bench.py --quick --steps 50 2>/dev/null | show "P16 224 B64"
