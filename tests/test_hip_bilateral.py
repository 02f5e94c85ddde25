"""Bilateral-solver refinement on the device against the reference's own outputs (tests/golden/bilateral.npz) and the
CPU oracle on fresh inputs (including exactly-grey pixels, which sit on lattice bin boundaries)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bilateral_oracle as B  # noqa: E402  (checker only)
from selfmask_amd.bilateral_solver import bilateral_solver_output, bilateral_solver_output_device  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bilateral.npz")
DEV = "cuda:0"


def test_matches_reference_outputs():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        img, tgt = g[f"img_{i}"], g[f"target_{i}"]
        soft, binary, info = bilateral_solver_output_device(torch.from_numpy(img).to(DEV), torch.from_numpy(tgt).to(DEV),
                                                            return_info=True)
        info = info.cpu().numpy()
        assert info[0] == int(g[f"nvert_{i}"]), (info, g[f"nvert_{i}"])
        d = np.abs(soft.cpu().numpy() - g[f"soft_{i}"]).max()
        print(f"\ncase {i}: V={info[0]} cg_iters={info[1]} components={info[2]} max|soft-ref|={d:.2e}")
        assert d <= 1e-9
        assert np.array_equal(binary.cpu().numpy().astype(bool), g[f"binary_{i}"])


@pytest.mark.parametrize("h,w,seed", [(97, 131, 1), (224, 224, 2), (300, 400, 3)])
def test_matches_oracle_on_fresh_scenes(h, w, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[:h, :w]
    img = np.clip(100 + 60 * np.sin(xx / 19.0) + 40 * np.cos(yy / 23.0) + rng.standard_normal((h, w)) * 10, 0, 255)
    img = np.repeat(img[..., None], 3, -1).astype(np.uint8)  # grey image: every pixel on the chroma bin boundary
    img[h // 4:h // 2, w // 3:w // 2] = rng.integers(0, 256, size=3, dtype=np.uint8)
    blob = (((yy - h * .5) / (h * .25)) ** 2 + ((xx - w * .5) / (w * .2)) ** 2) <= 1
    ring = blob & ~((((yy - h * .5) / (h * .1)) ** 2 + ((xx - w * .5) / (w * .08)) ** 2) <= 1)  # a hole to fill
    tgt = np.clip(0.2 + 0.65 * ring + rng.standard_normal((h, w)) * 0.15, 0, 1)
    soft, binary = bilateral_solver_output(img, tgt, device=DEV)
    rs, rb, grid = B.bilateral_solver_output(img, tgt)
    assert np.abs(soft - rs).max() <= 1e-9
    assert np.array_equal(binary, rb)


def test_degenerate_targets():
    img = np.full((64, 64, 3), 90, np.uint8)
    for tgt in (np.zeros((64, 64)), np.ones((64, 64))):
        soft, binary = bilateral_solver_output(img, tgt, device=DEV)
        rs, rb, _ = B.bilateral_solver_output(img, tgt)
        assert np.abs(soft - rs).max() <= 1e-12 and np.array_equal(binary, rb)
