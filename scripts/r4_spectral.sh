#!/bin/bash
# round 4: spectral clusterer / voting tests + the pseudo_masks leg
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_spectral.py tests/test_hip_cluster.py tests/test_hip_voting.py tests/test_hip_pipeline.py -x -q 2>&1 | tee gpurun_out/r4/spectral_tests.log | tail -30 && \
timeout -k 10 300 python3 bench.py --only-leg pseudo_masks --no-cpu-baseline | tee gpurun_out/r4/leg_pseudo_masks.json
