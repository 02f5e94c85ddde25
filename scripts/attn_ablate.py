#!/usr/bin/env python3
"""Timing of the F16X2 attention kernel on the encoder / decoder shapes; run once per SM_ATTN_ABLATE value
(0 whole, 1 = staging only, 2 = only the first chunk staged: compute + one staging)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
# the ablation switches are compiled into the tuning build only (python salient-object-detection_amd/build.py --tuning)
os.environ.setdefault("SM_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))
from selfmask_amd import ops, _native as N

def t(fn, it=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

print("SM_ATTN_ABLATE =", os.environ.get("SM_ATTN_ABLATE"))
for name, B, nq, nk in [("enc", 64, 197, 197), ("dec_self", 64, 20, 20), ("dec_cross", 64, 20, 196)]:
    qkv = torch.randn(B, max(nq, nk), 3, 6, 64, device="cuda")
    s = ops.split_f16x2(qkv.view(B, -1, 3 * 384)).view(qkv.shape)
    q, k, v = s[:, :nq, 0], s[:, :nk, 1], s[:, :nk, 2]
    o = torch.empty(B, nq, 384, device="cuda")
    a = N.AttnArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    a.sQb, a.sQr, a.sKb, a.sKr, a.sVb, a.sVr = q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1)
    a.sOb, a.sOr = o.stride(0), o.stride(1)
    a.batch, a.heads, a.n_q, a.n_k, a.scale, a.out_f16x2 = B, 6, nq, nk, 0.125, 1
    lib = N.load()
    st = torch.cuda.current_stream().cuda_stream
    print(f"{name:10s} {t(lambda: N.check(lib.sm_attention_f16x2(a, st))):7.1f} us")
