#!/bin/bash
# round 4: does the bilateral solver of one batch overlap with the forward of the next?  refine_384 leg by streams in flight
mkdir -p gpurun_out/r4
for s in 1 2 3 4; do
  python3 bench.py --only-leg refine_384 --no-cpu-baseline --streams $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())['refine_384']
print('streams $s: refined', d['value'], 'images/s; plain', d['without_refinement_images_per_sec'], '; ratio', d['refined_over_plain'], '; solver alone ms/batch', d['solver_alone']['ms_per_batch'])"
done | tee gpurun_out/r4/refine_streams.log
