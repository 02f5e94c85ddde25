#!/bin/bash
# round-3 experiment 8: encoder pre-norms folded into the GEMMs around them (sm_weights.ln_fold): parity, then A/B in the pipeline
O=gpurun_out/r3m; mkdir -p $O
python -m pytest tests/test_hip_gemm_w16.py tests/test_hip_qkv_attention.py -x -q -m gpu > $O/tests_unit.log 2>&1; echo "unit rc=$?"; tail -3 $O/tests_unit.log
python -m pytest tests/test_hip_forward.py tests/test_hip_evaluator.py tests/test_hip_inference.py -x -q -m gpu -s > $O/tests_fwd.log 2>&1; echo "fwd rc=$?"; tail -3 $O/tests_fwd.log
grep "hip-ref32" $O/tests_fwd.log | grep "w16\]" | head -12
one() { python bench.py --quick --steps 80 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "fold      "; one
  echo -n "LN kernels"; SM_LN_FOLD=0 one
done | tee $O/ln_fold_ab.log
for B in 1 8; do
  echo -n "B=$B fold       "; one --batch $B --streams 1 --steps 200 --warmup 20 --forward-only
  echo -n "B=$B LN kernels "; SM_LN_FOLD=0 one --batch $B --streams 1 --steps 200 --warmup 20 --forward-only
done | tee -a $O/ln_fold_ab.log
