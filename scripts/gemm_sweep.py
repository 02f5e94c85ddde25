#!/usr/bin/env python3
"""Time every GEMM shape of the B=64 ViT-S/16 forward with every workgroup tile (HIP events, current stream)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
from selfmask_amd import ops, _native as N

def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 197
shapes = [("qkv", M, 1152, 384, N.EPI_BIAS), ("proj", M, 384, 384, N.EPI_RESIDUAL), ("fc1", M, 1536, 384, N.EPI_GELU),
          ("fc2", M, 384, 1536, N.EPI_RESIDUAL), ("dec_kv", B * 196, 768, 384, N.EPI_BIAS),
          ("dec_small", B * 20, 384, 384, N.EPI_BIAS), ("dec_qk", B * 20, 768, 384, N.EPI_BIAS),
          ("dec_lin1", B * 20, 1536, 384, N.EPI_RELU), ("dec_lin2", B * 20, 384, 1536, N.EPI_RESIDUAL),
          ("obj", B * 120, 384, 384, N.EPI_RELU), ("patch", B * 196, 384, 768, N.EPI_BIAS)]
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    c = torch.empty(m, n, device="cuda"); r = torch.randn(m, n, device="cuda")
    res = []
    for tile in [(128, 128), (128, 64), (64, 64)]:
        us = t(lambda: ops.gemm(a, w, b, epilogue=epi, residual=r if epi == N.EPI_RESIDUAL else None, out=c, tile=tile))
        res.append(f"{tile[0]}x{tile[1]}: {us:7.1f} us {2.0*m*n*k/us/1e6:6.1f} TF")
    print(f"{name:10s} M={m:6d} N={n:5d} K={k:5d} | " + " | ".join(res))
