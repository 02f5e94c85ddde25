#!/bin/bash
# full GPU suite + quick bench after the evaluator / im2col / up-sample changes
set -e
mkdir -p gpurun_out/r3k
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r3k/tests.log
python bench.py --quick --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quick full        ', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3k/tests.log
python bench.py --quick --forward-only --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('quick forward-only', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r3k/tests.log
