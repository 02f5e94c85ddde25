#!/bin/bash
# round 4: first run of the spectral clusterer on the device
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_hip_spectral.py tests/test_hip_cluster.py -x -q 2>&1 | tee gpurun_out/r4/spectral_tests.log | tail -40
