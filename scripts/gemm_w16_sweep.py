#!/usr/bin/env python3
"""Encoder GEMM shapes (B=64, ViT-S/16: M = 12608): every W16 variant against the two-accumulator F16X2 kernel,
interleaved rounds in one process (median of rounds).  Prints us per launch and issued-MFMA TFLOP/s."""
import os, sys, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))  # the variant knobs live in the tuning build
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
from selfmask_amd import ops, _native as N

dev = "cuda:0"
M = int(os.environ.get("M", 12608))
SHAPES = [("qkv", 1152, 384, N.EPI_BIAS, True, False), ("proj", 384, 384, N.EPI_RESIDUAL, False, True),
          ("fc1", 1536, 384, N.EPI_GELU, True, False), ("fc2", 384, 1536, N.EPI_RESIDUAL, False, True),
          ("kv", 4608, 384, N.EPI_BIAS, True, False)]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "2,7,30,31,32,33,34,35").split(",")]
ROUNDS, ITERS = 7, 20


def timeit(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / ITERS


for name, Nn, K, epi, osplit, res in SHAPES:
    g = torch.Generator().manual_seed(1)
    zero = os.environ.get("ZERO") == "1"  # all-zero operands: same instruction stream, far less switching power
    a = ops.split_f16x2((torch.randn(M, K, generator=g) * (0.0 if zero else 1.0)).to(dev))
    w = (torch.randn(Nn, K, generator=g) * 0.03).to(dev)
    if zero:
        w = w * 0.0 + 1e-30
    b = torch.randn(Nn, generator=g).to(dev)
    w_s, (w16, ws) = ops.split_f16x2(w), ops.split_w16(w)
    out = torch.empty(1, M, Nn, device=dev)
    r = torch.randn(M, Nn, device=dev) if res else None
    cands = {"f16x2 128x128": lambda: ops.gemm_f16x2(a, w_s, b, epilogue=epi, residual=r, tile=(128, 128), out=out, out_f16x2=osplit)}
    for v in VARIANTS:
        cands[f"w16 v{v}"] = (lambda v=v: ops.gemm_w16(a, w16, ws, b, epilogue=epi, residual=r, variant=v, out=out, out_f16x2=osplit))
    times = {k: [] for k in cands}
    for _ in range(ROUNDS):
        for k, fn in cands.items():
            times[k].append(timeit(fn))
    flops = 2.0 * M * Nn * K
    print(f"{name}: M={M} N={Nn} K={K}")
    for k, t in times.items():
        med = statistics.median(t)
        print(f"   {k:16s} {med:7.1f} us (min {min(t):6.1f})   {flops / med / 1e6:6.1f} TFLOP/s alg, {3 * flops / med / 1e6:7.1f} issued")
