#!/bin/bash
# stream priorities in the ring (SM_STREAM_PRIORITIES), three alternations; then the default bench line
mkdir -p gpurun_out/r3l
for rep in 1 2 3; do
  for p in "" "-1,0,0" "-1,-1,0" "0,0,-1"; do
    SM_STREAM_PRIORITIES=$p python bench.py --quick --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('priorities [$p]', d['value'], d['ms_per_step'])"
  done
done | tee gpurun_out/r3l/stream_priorities.log
python bench.py > gpurun_out/r3l/r03_bench_default.json 2> gpurun_out/r3l/err.log; tail -c 200 gpurun_out/r3l/r03_bench_default.json
