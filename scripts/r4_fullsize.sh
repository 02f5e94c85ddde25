#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_forward.py tests/test_hip_bilateral.py -x -q -s -k "other_bench_shapes or bench_batch_384" 2>&1 | tee gpurun_out/r4/fullsize_tests.log | tail -15
