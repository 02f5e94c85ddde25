#!/bin/bash
# round 4: ViT-S/8 224^2 (N = 785) by batch size: the attention launch has 7 groups x 6 heads x B workgroups on 512 slots - at B = 16
# that is 1.31 rounds of work in 2; where does the tail go in the three-stream pipeline?
mkdir -p gpurun_out/r4
for b in 16 24 32; do
  for s in 1 3; do
    python3 bench.py --quick --patch 8 --batch $b --streams $s --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
att=[v for k,v in {**{d['roofline']['kernel']:d['roofline']}, **d['roofline_other_kernels']}.items() if 'attention_f16x2' in k]
print('batch $b streams $s', d['value'], 'images/s', d['ms_per_step'], 'ms/step; attention_f16x2', att[0]['avg_launch_us'] if att else None, 'us/launch', round(att[0]['avg_launch_us']/$b,3) if att else None, 'us/image')"
  done
done | tee gpurun_out/r4/s8_batches.log
