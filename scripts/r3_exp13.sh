#!/bin/bash
# per-stream grow-only workspace: parity (forward / evaluator / inference / streams / graphs), then the native-resolution legs three times
mkdir -p gpurun_out/r3m
python -m pytest tests/test_hip_forward.py tests/test_hip_evaluator.py tests/test_hip_inference.py tests/test_hip_streams.py tests/test_hip_graphs.py -x -q 2>&1 | tail -4
python scripts/e2e_workers.py 15 15 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3m/e2e_native.log
