#!/bin/bash
for rep in 1 2; do for t in 1 2 3 4; do echo "pack threads $t"; SM_PACK_THREADS=$t python scripts/e2e_workers.py 15 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/pack_threads.log
