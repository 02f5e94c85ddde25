#!/bin/bash
mkdir -p gpurun_out/r3p
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python scripts/host_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3p/host_profile.log; grep "==" gpurun_out/r3p/host_profile.log
bash scripts/profile_bench.sh r03_b1 --steps 30 --warmup 6 --streams 1 --no-graph --batch 1 > gpurun_out/profile_b1.log 2>&1; python scripts/trace_summary.py gpurun_out/r03_b1 > gpurun_out/r3p/r03_forward_breakdown_b1.txt 2>&1; head -8 gpurun_out/r3p/r03_forward_breakdown_b1.txt
