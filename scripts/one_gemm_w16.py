#!/usr/bin/env python3
"""One W16 GEMM shape / variant, a few launches without cache flushes: for `rocprofv3 --pmc` passes (SQ counters).
usage: one_gemm_w16.py M N K variant [epilogue: bias|gelu|residual] [out_f16x2: 0|1]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))  # the variant knobs live in the tuning build
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
from selfmask_amd import ops, _native as N
M, Nn, K, variant = (int(v) for v in sys.argv[1:5])
epi = {"bias": N.EPI_BIAS, "gelu": N.EPI_GELU, "residual": N.EPI_RESIDUAL}[sys.argv[5] if len(sys.argv) > 5 else "bias"]
osplit = len(sys.argv) > 6 and sys.argv[6] == "1"
a = ops.split_f16x2(torch.randn(M, K, device="cuda"))
w16, ws = ops.split_w16(torch.randn(Nn, K, device="cuda") * 0.05)
b = torch.randn(Nn, device="cuda"); r = torch.randn(M, Nn, device="cuda") if epi == N.EPI_RESIDUAL else None
c = torch.empty(1, M, Nn, device="cuda")
for _ in range(10):
    ops.gemm_w16(a, w16, ws, b, epilogue=epi, residual=r, variant=variant, out=c, out_f16x2=osplit)
torch.cuda.synchronize()
