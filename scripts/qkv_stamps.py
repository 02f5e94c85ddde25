#!/usr/bin/env python3
"""Where a workgroup of the fused QKV+attention kernel spends its life: in-kernel s_memtime stamps (tuning build only,
`build.py --tuning`), median over workgroups, B = 64, N = 197, random operands, after a warm-up of back-to-back launches.
Stamps: 0 start | 1 first stage landed | 2 projection loop done | 3 ring drained + barrier | 4 Q/K/V^T converted + barrier |
5 attention loop done | 6 output stored."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
from selfmask_amd import ops, _native as N

dev, B, n = "cuda:0", 64, 197
g = torch.Generator().manual_seed(0)
xs = ops.split_f16x2(torch.randn(B * n, 384, generator=g).to(dev))
w16, ws = ops.split_w16((torch.randn(1152, 384, generator=g) * 0.05).to(dev))
b = (torch.randn(1152, generator=g) * 0.1).to(dev)
o = torch.empty(B * n, 384, device=dev)
lib = N.load()
a = N.QkvAttnArgs()
a.Xn, a.Wqkv, a.bias, a.O = xs.data_ptr(), w16.data_ptr(), b.data_ptr(), o.data_ptr()
a.ldx, a.ldo, a.B, a.N, a.w_scale, a.scale, a.out_f16x2 = 384, 384, B, n, ws, 0.125, 1
st = torch.cuda.current_stream().cuda_stream
for _ in range(300):
    N.check(lib.sm_qkv_attention_w16(a, st))
torch.cuda.synchronize()
WAVES, WG = 7, B * 6
buf = (ctypes.c_ulonglong * (WG * WAVES * 8))()
lib.sm_qkv_stamps.restype = ctypes.c_int
assert lib.sm_qkv_stamps(buf, WG * WAVES * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(WG, WAVES, 8).astype(np.int64)
d = np.diff(t[:, :, :7], axis=2)
names = ["prologue (first stage lands)", "projection loop (12 stages)", "drain + barrier", "convert Q/K/V^T + barrier",
         "attention loop (7 steps)", "normalise + store"]
tot = t[:, :, 6] - t[:, :, 0]
print(f"kernel {os.environ.get('SM_QKV_RING', 'm16x2')}: s_memtime ticks, median over {WG} workgroups")
print(" phases: " + " | ".join(names))
for w in range(WAVES):
    print(f" wave {w}: " + "  ".join(f"{np.median(d[:, w, i]):8.0f}" for i in range(6)) + f"   total {np.median(tot[:, w]):8.0f}")
print(" all   : " + "  ".join(f"{np.median(d[:, :, i]):8.0f}" for i in range(6)) + f"   total {np.median(tot):8.0f}")
wg_life = t[:, :, 6].max(1) - t[:, :, 0].min(1)
print(f" workgroup life (first start -> last end): median {np.median(wg_life):.0f}, p10 {np.percentile(wg_life, 10):.0f}, p90 {np.percentile(wg_life, 90):.0f} ticks")
