#!/bin/bash
# the clusterer by number of points (one workgroup per image; 16 images): 784 = LDS-resident, 1024 / 1444 / 1936 = graph in the LDS and
# 8 / 4 / 2 columns staged per pass, 3136 = graph and blocks in memory
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r4
python - > gpurun_out/r4/spectral_sizes.log 2>&1 <<'PY'
import os, sys
sys.path[:0] = ["salient-object-detection_amd", ".", "tests"]
import numpy as np, torch
from selfmask_amd import voting as VT
from test_oracle_spectral import scene
for g in (28, 32, 38, 44, 50, 56):
    B = 16
    x = torch.from_numpy(np.stack([scene(g, 3 + s % 2, 100 + s)[0] for s in range(B)])).cuda()
    VT.spectral_cluster(x, (2, 3, 4)); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        VT.spectral_cluster(x, (2, 3, 4))
    e1.record(); torch.cuda.synchronize()
    _, det = VT.spectral_cluster(x, (2, 3, 4), return_details=True)
    info = det["info"].cpu().numpy()
    ms = e0.elapsed_time(e1) / 3
    print(f"n = {g * g:5d}: {ms:8.3f} ms per batch of {B}; {info[:, 1].mean():6.1f} block mat-vecs -> {ms * 1e3 / info[:, 1].mean():7.2f} us per mat-vec, "
          f"{ms * 1e6 / info[:, 1].mean() / (g * g):6.2f} ns per row and mat-vec; converged {int(info[:, 2].sum())}/{B}", flush=True)
PY
cat gpurun_out/r4/spectral_sizes.log
