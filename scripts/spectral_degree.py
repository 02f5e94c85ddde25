#!/usr/bin/env python3
"""Where the eigen-solver's time goes: the same features with Chebyshev filters of different degrees (fewer, longer filters = fewer
orthonormalisations / Rayleigh-Ritz steps for about the same number of mat-vecs).  usage: spectral_degree.py [grid side g | bench] [batch]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
from selfmask_amd import voting as VT
from test_oracle_spectral import scene
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if len(sys.argv) > 1 and sys.argv[1] == "bench":  # the bench leg's own features: random-weight ViT-S/16 tokens of synthetic images, x2
    import bench
    w = bench.Workload(torch.device("cuda:0"), 16, 224, B, streams=1, forward_only=True, graph=False)
    tok = w.model(w.x, encoder_only=True)["patch_tokens"]
    gh, gw = tok.shape[1:3]
    g = 2 * gh
    x = VT.upsample_tokens_aligned(tok.reshape(B, gh * gw, 384), gh, gw, 2).reshape(B, 4 * gh * gw, 384)
else:
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    x = torch.from_numpy(np.stack([scene(g, 3 + s % 2, 100 + s)[0] for s in range(B)])).cuda()
for degree in (12, 16, 24, 32, 48, 64):
    VT.spectral_cluster(x, (2, 3, 4), degree=degree)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        VT.spectral_cluster(x, (2, 3, 4), degree=degree)
    e1.record(); torch.cuda.synchronize()
    _, det = VT.spectral_cluster(x, (2, 3, 4), degree=degree, return_details=True)
    info = det["info"].cpu().numpy()
    print(f"n = {g * g}, batch {B}, degree {degree:3d}: {e0.elapsed_time(e1) / 5:7.3f} ms per batch; filters {info[:, 0].mean():5.2f}, block mat-vecs {info[:, 1].mean():6.1f}, "
          f"converged {int(info[:, 2].sum())}/{B}, max residual {float(det['residuals'].max()):.1e}")
