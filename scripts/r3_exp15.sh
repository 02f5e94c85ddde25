#!/bin/bash
mkdir -p gpurun_out/r3o
python -m pytest tests/test_hip_inference.py -x -q 2>&1 | tail -3
python scripts/host_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3o/host_profile.log; grep "==" gpurun_out/r3o/host_profile.log
python bench.py > gpurun_out/r3o/r03_bench_default.json 2> gpurun_out/r3o/err.log; python - <<'PY'
import json
d = json.load(open("gpurun_out/r3o/r03_bench_default.json"))
print("bench", d["value"], "serving", d["serving"], "native", d["end_to_end"]["native_resolution"]["bucketed_batch16_images_per_sec"], "e2e", d["end_to_end"]["end_to_end_images_per_sec"])
PY
