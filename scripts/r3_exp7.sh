#!/bin/bash
# round-3 experiment 7: ring-of-three GEMMs (proj, fc2, patch): second half of the waves issues its LDS-DMA pieces after its MFMAs
L=$PWD/salient-object-detection_amd/lib/libselfmask_hip_late.so
python - <<PY
import os, sys
os.environ["SM_HIP_LIB"] = "$L"
sys.path[:0] = ["salient-object-detection_amd", "."]
import torch
from selfmask_amd import ops, _native as N
g = torch.Generator().manual_seed(1)
for (M, Nn, K, epi) in ((12608, 384, 1536, N.EPI_RESIDUAL), (12608, 384, 384, N.EPI_RESIDUAL), (600, 384, 1536, N.EPI_BIAS)):
    a, w, b = torch.randn(M, K, generator=g), torch.randn(Nn, K, generator=g) * 0.05, torch.randn(Nn, generator=g)
    r = torch.randn(M, Nn, generator=g) if epi == N.EPI_RESIDUAL else None
    w16, ws = ops.split_w16(w.cuda())
    c = ops.gemm_w16(ops.split_f16x2(a.cuda()), w16, ws, b.cuda(), epilogue=epi, residual=None if r is None else r.cuda(), variant=47)
    ref = a.double() @ w.double().T + b.double() + (r.double() if r is not None else 0)
    print("late-half v47", M, Nn, K, "max err", (c.double().cpu() - ref).abs().max().item() / ref.abs().max().item())
PY
one() { python bench.py --quick --steps 80 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline_other_kernels']; print(d['value'], d['ms_per_step'], [ (n[19:40], v['avg_launch_us']) for n,v in k.items() if '256, 128' in n])"; }
for i in 1 2 3; do
  echo -n "base  "; one
  echo -n "late  "; SM_HIP_LIB=$L one
done
