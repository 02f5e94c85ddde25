"""Time sm_evaluate_masks_f32 alone on the bench workload's shapes (B = 64, nq = 20, GT 300-400 px).

    python scripts/eval_bench.py [mask_side ...]          (default 28 56 48: ViT-S/16 224, ViT-S/8 224, ViT-S/16 384)
With the tuning library (SM_HIP_LIB=.../libselfmask_hip_tuning.so) SM_EVAL_BAND_MIN=0 forces the raster walk."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "salient-object-detection_amd"))
from selfmask_amd import ops  # noqa: E402


def main():
    dev = "cuda:0"
    sides = [int(v) for v in sys.argv[1:]] or [28, 56, 48]
    rng = np.random.Generator(np.random.PCG64(99))
    B, nq = 64, 20
    gts = []
    for _ in range(B):
        h, w = (int(v) for v in rng.integers(300, 401, size=2))
        yy, xx = np.mgrid[:h, :w]
        gts.append(torch.from_numpy(((((yy - h * rng.uniform(.3, .7)) / (h * rng.uniform(.1, .3))) ** 2 +
                                      ((xx - w * rng.uniform(.3, .7)) / (w * rng.uniform(.1, .3))) ** 2) <= 1).astype(np.uint8)))
    gb = ops.GtBatch(gts, dev)
    npx = sum(h * w for (h, w) in gb.shapes)
    for side in sides:
        torch.manual_seed(side)
        mp = torch.sigmoid(torch.randn(B, nq, side, side, device=dev) * 3)
        ob = torch.rand(B, nq, device=dev)
        for _ in range(3):
            rows = ops.evaluate_masks(mp, ob, gb, scale=0.0)
        torch.cuda.synchronize()
        n = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            rows = ops.evaluate_masks(mp, ob, gb, scale=0.0)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        # algorithmic bytes: the GT once per pass that needs it (query, metrics x 2 selections) + the mask planes
        print(f"mask {side:3d}^2  B={B} nq={nq}  {npx / B / 1e3:.0f}k px/image: {us:8.1f} us per batch  "
              f"({B / us * 1e6:,.0f} images/s, {npx * (1 + nq) / us / 1e3:.1f} G query-pixels/s)  rows checksum {float(rows.double().nan_to_num().sum()):.6f}")


if __name__ == "__main__":
    main()
