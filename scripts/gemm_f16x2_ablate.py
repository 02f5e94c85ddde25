#!/usr/bin/env python3
"""Where the split-operand GEMM's time goes: each tile shape timed whole and with one part removed (SM_F16X2_ABLATE:
2 = no MFMA / LDS reads, 3 = no LDS-DMA inside the loop, 4 = no epilogue).  Timing only - ablated runs compute garbage."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
# the ablation switches are compiled into the tuning build only (python salient-object-detection_amd/build.py --tuning)
os.environ.setdefault("SM_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))
from selfmask_amd import ops, _native as N

def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it

B = 64; M = B * 197
shapes = [("qkv", M, 1152, 384, N.EPI_BIAS, True), ("proj", M, 384, 384, N.EPI_RESIDUAL, False),
          ("fc1", M, 1536, 384, N.EPI_GELU, True), ("fc2", M, 384, 1536, N.EPI_RESIDUAL, False),
          ("dec_q", B * 20, 384, 384, N.EPI_BIAS, True), ("dec_l1", B * 20, 1536, 384, N.EPI_RELU, True),
          ("kv_all", B * 196, 4608, 384, N.EPI_BIAS, True), ("mask_mlp", B * 120, 384, 384, N.EPI_RELU, True)]
torch.manual_seed(0)
for name, m, n, k, epi, osplit in shapes:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    r = torch.randn(m, n, device="cuda") if epi == N.EPI_RESIDUAL else None
    a_s, w_s = ops.split_f16x2(a), ops.split_f16x2(w)
    c_out = torch.empty(1, m, n, device="cuda")
    for tile in [(256, 128), (128, 128), (128, 64), (64, 64)]:
        for nst in ["2"]:
            os.environ["SM_F16X2_NST"] = nst
            line = f"{name:5s} {tile[0]}x{tile[1]} nst={nst}:"
            for ab in ["0", "2", "3", "4", "5", "6"]:
                if ab == "0": os.environ.pop("SM_F16X2_ABLATE", None)
                else: os.environ["SM_F16X2_ABLATE"] = ab
                us = t(lambda: ops.gemm_f16x2(a_s, w_s, b, epilogue=epi, residual=r, tile=tile, out=c_out, out_f16x2=osplit))
                line += f"  {['full','noMFMA','noDMA','noEPI','mfma+lds','nobarrier'][['0','2','3','4','5','6'].index(ab)]} {us:6.1f}"
            os.environ.pop("SM_F16X2_ABLATE", None)
            print(line, flush=True)
