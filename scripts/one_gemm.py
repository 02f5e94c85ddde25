#!/usr/bin/env python3
"""One split-operand GEMM shape, default tile, a few launches: for `rocprofv3 --pmc` passes (HBM bytes of one shape).
usage: one_gemm.py M N K [epilogue: bias|gelu|residual] [out_f16x2: 0|1]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import ctypes
import torch
from selfmask_amd import ops, _native as N
M, Nn, K = (int(v) for v in sys.argv[1:4])
epi = {"bias": N.EPI_BIAS, "gelu": N.EPI_GELU, "residual": N.EPI_RESIDUAL}[sys.argv[4] if len(sys.argv) > 4 else "bias"]
osplit = len(sys.argv) > 5 and sys.argv[5] == "1"
a = ops.split_f16x2(torch.randn(M, K, device="cuda")); w = ops.split_f16x2(torch.randn(Nn, K, device="cuda") * 0.05)
b = torch.randn(Nn, device="cuda"); r = torch.randn(M, Nn, device="cuda") if epi == N.EPI_RESIDUAL else None
g = N.GemmArgs(); g.M, g.N, g.K, g.batch = M, Nn, K, 1
bm, bn, nst = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
N.load().sm_gemm_f16x2_pick_tile(g, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(nst))
tile = tuple(int(v) for v in os.environ.get("TILE", f"{bm.value}x{bn.value}").split("x"))
c = torch.empty(1, M, Nn, device="cuda")
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for _ in range(6):
    big.fill_(1)  # flush L2 / Infinity Cache between launches
    ops.gemm_f16x2(a, w, b, epilogue=epi, residual=r, tile=tile, out=c, out_f16x2=osplit)
torch.cuda.synchronize()
print("tile", tile)
