#!/bin/bash
mkdir -p gpurun_out/r3n
python -m pytest tests/test_hip_pipeline.py tests/test_hip_evaluator.py tests/test_hip_inference.py -x -q 2>&1 | tail -3
python scripts/e2e_workers.py 15 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3n/e2e_native.log
python bench.py > gpurun_out/r3n/r03_bench_default.json 2> gpurun_out/r3n/err.log; tail -c 200 gpurun_out/r3n/r03_bench_default.json
