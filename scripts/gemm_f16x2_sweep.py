#!/usr/bin/env python3
"""Correctness (vs fp64) and speed of the split-operand f16 GEMM against the exact-fp32 MFMA GEMM."""
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

B = 64; M = B * 197
shapes = [("qkv", M, 1152, 384, N.EPI_BIAS), ("proj", M, 384, 384, N.EPI_RESIDUAL), ("fc1", M, 1536, 384, N.EPI_GELU),
          ("fc2", M, 384, 1536, N.EPI_RESIDUAL), ("dec_small", B * 20, 384, 384, N.EPI_BIAS)]
print("SM_F16X2_NST =", os.environ.get("SM_F16X2_NST"))
torch.manual_seed(0)
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    r = torch.randn(m, n, device="cuda") if epi == N.EPI_RESIDUAL else None
    a_s, w_s = ops.split_f16x2(a), ops.split_f16x2(w)
    ref = a.double() @ w.double().T + b.double()
    if epi == N.EPI_GELU: ref = torch.nn.functional.gelu(ref)
    if r is not None: ref = ref + r.double()
    c32 = ops.gemm(a, w, b, epilogue=epi, residual=r.clone() if r is not None else None)
    line = f"{name:9s} M={m} N={n} K={k} | fp32-mfma err {(c32.double()-ref).abs().max().item():.2e}"
    for tile in [(256, 128), (256, 64), (128, 128), (128, 64), (64, 64)]:
        c = ops.gemm_f16x2(a_s, w_s, b, epilogue=epi, residual=r, tile=tile)
        err = (c.double() - ref).abs().max().item()
        c_out = torch.empty(m, n, device="cuda")
        us = t(lambda: ops.gemm_f16x2(a_s, w_s, b, epilogue=epi, residual=r, tile=tile, out=c_out))
        line += f" | {tile[0]}x{tile[1]}: err {err:.2e} {us:6.1f} us {2.0*m*n*k/us/1e6:6.1f} TF"
    print(line)
# F16X2 output round trip: fc1 (GELU) -> split output -> fc2 consumes it
a = torch.randn(M, 384, device="cuda"); w1 = torch.randn(1536, 384, device="cuda") * 0.05; w2 = torch.randn(384, 1536, device="cuda") * 0.03
hid = ops.gemm_f16x2(ops.split_f16x2(a), ops.split_f16x2(w1), None, epilogue=N.EPI_GELU, out_f16x2=True)
out = ops.gemm_f16x2(hid, ops.split_f16x2(w2), None)
ref = torch.nn.functional.gelu(a.double() @ w1.double().T) @ w2.double().T
ref32 = torch.nn.functional.gelu(a @ w1.T) @ w2.T
print("chain fc1->gelu->fc2: err vs fp64", (out.double() - ref).abs().max().item(), " torch-fp32 err", (ref32.double() - ref).abs().max().item())
