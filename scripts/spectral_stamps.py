#!/usr/bin/env python3
"""Phase cycles of the spectral eigen-solver (experiment build: build.py --variant=spstamps -DSM_SPECTRAL_STAMPS; the residual output then
carries shader-clock cycles of thread 0: filter steps, Cholesky-QR passes, symmetric mat-vecs, Rayleigh-Ritz).  usage: spectral_stamps.py [g]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SM_HIP_LIB", os.path.join(REPO, "salient-object-detection_amd", "lib", "libselfmask_hip_spstamps.so"))
sys.path[:0] = [os.path.join(REPO, "salient-object-detection_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
from selfmask_amd import voting as VT
from test_oracle_spectral import scene
g = int(sys.argv[1]) if len(sys.argv) > 1 else 28
x = torch.from_numpy(np.stack([scene(g, 3 + s % 2, 100 + s)[0] for s in range(4)])).cuda()
VT.spectral_cluster(x, (2, 3, 4))
_, det = VT.spectral_cluster(x, (2, 3, 4), return_details=True)
st, info = det["residuals"].cpu().numpy(), det["info"].cpu().numpy()
for b in range(4):
    d = det["embedding"][b].cpu().numpy().reshape(-1)[:16]
    print(f"    list groups of four: {d[11]:.0f} of {d[12]:.0f} that fit")
    f, c, a, r = st[b]
    tot = f + c + a + r
    print(f"[info {info[b, 3]:#x}] n = {g * g}, image {b}: {info[b, 0]} filters, {info[b, 1]} block mat-vecs; cycles (100 MHz counter units x clock ratio): filter {f:.0f} ({f / tot:.0%}; "
          f"{f / max(1, info[b, 1] - info[b, 0] - 1):.0f} per step), Cholesky-QR {c:.0f} ({c / tot:.0%}; {c / (3 * (info[b, 0] + 1)):.0f} per pass), "
          f"symmetric mat-vec {a:.0f} ({a / tot:.0%}), Rayleigh-Ritz {r:.0f} ({r / tot:.0%}; {r / (info[b, 0] + 1):.0f} per call)")
