#!/usr/bin/env python3
"""Fused QKV+attention kernel vs the two launches it replaces (W16 qkv GEMM + F16X2 attention), B = 64, N = 197, one stream,
median of interleaved rounds.  Run once per SM_QKV_RING value (2, 3, 6): the ring depth is read once per process."""
import os, sys, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))  # the variant knobs live in the tuning build
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
from selfmask_amd import ops, _native as N

dev, B, n = "cuda:0", 64, 197
g = torch.Generator().manual_seed(0)
xn = torch.randn(B * n, 384, generator=g).to(dev)
w = (torch.randn(1152, 384, generator=g) * 0.05).to(dev)
b = (torch.randn(1152, generator=g) * 0.1).to(dev)
xs = ops.split_f16x2(xn)
w16, ws = ops.split_w16(w)
o = torch.empty(B * n, 384, device=dev)
qkv = torch.empty(1, B * n, 1152, device=dev)
lib = N.load()
a = N.QkvAttnArgs()
a.Xn, a.Wqkv, a.bias, a.O = xs.data_ptr(), w16.data_ptr(), b.data_ptr(), o.data_ptr()
a.ldx, a.ldo, a.B, a.N, a.w_scale, a.scale, a.out_f16x2 = 384, 384, B, n, ws, 0.125, 1
at = N.AttnArgs()
at.Q, at.K, at.V, at.O = qkv.data_ptr(), qkv.data_ptr() + 384 * 4, qkv.data_ptr() + 768 * 4, o.data_ptr()
at.sQb = at.sKb = at.sVb = n * 1152; at.sQr = at.sKr = at.sVr = 1152; at.sOb, at.sOr = n * 384, 384
at.batch, at.heads, at.n_q, at.n_k, at.scale, at.out_f16x2 = B, 6, n, n, 0.125, 1
st = torch.cuda.current_stream().cuda_stream


def fused():
    N.check(lib.sm_qkv_attention_w16(a, st))


def unfused():
    ops.gemm_w16(xs, w16, ws, b, out=qkv, out_f16x2=True)
    N.check(lib.sm_attention_f16x2(at, st))


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


t = {"fused": [], "unfused (gemm + attention)": []}
for _ in range(9):
    t["fused"].append(timeit(fused)); t["unfused (gemm + attention)"].append(timeit(unfused))
flops = B * (2.0 * n * 384 * 1152 + 4.0 * n * n * 384)
for k, v in t.items():
    m = statistics.median(v)
    print(f"SM_QKV_RING={os.environ.get('SM_QKV_RING', '2')} {k:28s} {m:7.1f} us (min {min(v):6.1f})  {flops / m / 1e6:6.1f} TFLOP/s alg, {3 * flops / m / 1e6:6.1f} issued")
