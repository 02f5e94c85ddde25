#!/usr/bin/env python3
"""Prologue / K loop / epilogue of a tile of the 16x16x32-MFMA W16 GEMM: in-kernel s_memtime stamps (tuning build only,
`build.py --tuning`), median over the tiles of one launch after 200 back-to-back launches on random operands.
Stamps: 0 kernel entry (addresses set up) | 1 first stage landed | 2 K loop done, ring drained | 3 epilogue's stores retired."""
import ctypes, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so"))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO]
import torch
from selfmask_amd import ops, _native as N

lib = N.load()
lib.sm_gemm_stamps.restype = ctypes.c_int
M = 64 * 197
SHAPES = [("qkv  (bias, F16X2 out)", 1152, 384, N.EPI_BIAS, True, 41), ("proj (residual)", 384, 384, N.EPI_RESIDUAL, False, 41),
          ("fc1  (GELU, F16X2 out)", 1536, 384, N.EPI_GELU, True, 40), ("fc1  (GELU, F16X2 out)", 1536, 384, N.EPI_GELU, True, 41),
          ("fc1-shaped, bias only", 1536, 384, N.EPI_BIAS, True, 40), ("fc2  (residual)", 384, 1536, N.EPI_RESIDUAL, False, 41),
          ("fc2  (residual)", 384, 1536, N.EPI_RESIDUAL, False, 42), ("proj (residual)", 384, 384, N.EPI_RESIDUAL, False, 46),
          ("proj (residual)", 384, 384, N.EPI_RESIDUAL, False, 47), ("fc2  (residual)", 384, 1536, N.EPI_RESIDUAL, False, 47),
          ("proj (residual)", 384, 384, N.EPI_RESIDUAL, False, 48), ("fc2  (residual)", 384, 1536, N.EPI_RESIDUAL, False, 48),
          ("fc2  (residual)", 384, 1536, N.EPI_RESIDUAL, False, 46)]
TILE = {40: (256, 256), 41: (256, 128), 42: (128, 128), 46: (128, 384), 47: (256, 128), 48: (256, 128)}
print("variant / shape: tiles, then median ticks of the shader clock: prologue | K loop | epilogue | tile life   (launch us)")
for name, Nn, K, epi, osplit, variant in SHAPES:
    a = ops.split_f16x2(torch.randn(M, K, device="cuda"))
    w16, ws = ops.split_w16(torch.randn(Nn, K, device="cuda") * 0.05)
    b = torch.randn(Nn, device="cuda")
    r = torch.randn(M, Nn, device="cuda") if epi == N.EPI_RESIDUAL else None
    c = torch.empty(1, M, Nn, device="cuda")
    for _ in range(200):
        ops.gemm_w16(a, w16, ws, b, epilogue=epi, residual=r, variant=variant, out=c, out_f16x2=osplit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_w16(a, w16, ws, b, epilogue=epi, residual=r, variant=variant, out=c, out_f16x2=osplit)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    bm, bn = TILE[variant]
    nt = min(-(-M // bm) * -(-Nn // bn), 2048)
    buf = (ctypes.c_ulonglong * (nt * 4))()
    assert lib.sm_gemm_stamps(buf, nt * 4) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nt, 4).astype(np.int64)
    d = np.diff(t, axis=1)
    life = t[:, 3] - t[:, 0]
    span = t[:, 3].max() - t[:, 0].min()
    print(f" v{variant} {bm}x{bn} {name:26s} {nt:4d} tiles: {np.median(d[:, 0]):7.0f} | {np.median(d[:, 1]):7.0f} | {np.median(d[:, 2]):7.0f} | "
          f"{np.median(life):7.0f}   first entry -> last exit {span:7d} ticks   ({us:.1f} us)")
