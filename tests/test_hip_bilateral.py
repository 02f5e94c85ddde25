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


def test_batched_solve_equals_per_image_solves():
    """sm_bilateral_solver_batch_f64 (one workgroup per image in the long kernels): every image of a batch of different
    scenes - including an all-zero target - gives the bits of its own single solve, vertex / iteration counts included."""
    from selfmask_amd.bilateral_solver import bilateral_solver_batch_device
    h, w, n = 120, 152, 7
    rng = np.random.Generator(np.random.PCG64(11))
    yy, xx = np.mgrid[:h, :w]
    imgs, tgts = [], []
    for i in range(n):
        img = np.clip(np.stack([110 + 50 * np.sin(xx / (11.0 + 3 * i)), 90 + 60 * np.cos(yy / (9.0 + 2 * i)),
                                80 + 0.5 * xx + 0 * yy], -1) + rng.standard_normal((h, w, 3)) * (3 + 2 * i), 0, 255).astype(np.uint8)
        blob = (((yy - h * rng.uniform(.35, .65)) / (h * .25)) ** 2 + ((xx - w * rng.uniform(.35, .65)) / (w * .2)) ** 2) <= 1
        tgts.append(np.zeros((h, w)) if i == 3 else np.clip(0.15 + 0.7 * blob + rng.standard_normal((h, w)) * 0.1, 0, 1))
        imgs.append(img)
    I = torch.from_numpy(np.stack(imgs)).to(DEV)
    T = torch.from_numpy(np.stack(tgts)).to(DEV)
    soft, binary, info = bilateral_solver_batch_device(I, T, return_info=True)
    for i in range(n):
        s1, b1, i1 = bilateral_solver_output_device(I[i], T[i], return_info=True)
        assert torch.equal(soft[i], s1) and torch.equal(binary[i], b1) and torch.equal(info[i], i1), i
    rs, rb, _ = B.bilateral_solver_output(imgs[5], tgts[5])
    assert np.abs(soft[5].cpu().numpy() - rs).max() <= 1e-9 and np.array_equal(binary[5].cpu().numpy().astype(bool), rb)


def test_refinement_at_the_bench_batch_384_x_32():
    """BASELINE configs[2] at the size bench.py's refine_384 leg runs: 32 scenes of 384^2 in one batched solve.  Three of them equal
    their own single solves bit for bit (vertex / iteration counts included), one is checked against the CPU oracle, a second call
    reproduces the first."""
    from selfmask_amd.bilateral_solver import bilateral_solver_batch_device
    from selfmask_amd.datasets import synthetic_scene
    S, n = 384, 32
    rng = np.random.Generator(np.random.PCG64(77))
    scenes = [synthetic_scene(rng, S, S) for _ in range(n)]
    imgs = np.stack([im for im, _ in scenes])
    tgts = np.stack([np.clip(0.15 + 0.7 * g + rng.standard_normal((S, S)) * 0.1, 0, 1) for _, g in scenes])
    I, T = torch.from_numpy(imgs).to(DEV), torch.from_numpy(tgts).to(DEV)
    soft, binary, info = bilateral_solver_batch_device(I, T, return_info=True)
    soft2, binary2, info2 = bilateral_solver_batch_device(I, T, return_info=True)
    assert torch.equal(soft, soft2) and torch.equal(binary, binary2) and torch.equal(info, info2)
    for i in (0, 13, 31):
        s1, b1, i1 = bilateral_solver_output_device(I[i], T[i], return_info=True)
        assert torch.equal(soft[i], s1) and torch.equal(binary[i], b1) and torch.equal(info[i], i1), i
    rs, rb, _ = B.bilateral_solver_output(imgs[13], tgts[13])
    d = np.abs(soft[13].cpu().numpy() - rs).max()
    print(f"\n384^2 x 32: V={int(info[13, 0])} cg_iters={int(info[13, 1])} max|soft-oracle|={d:.2e}")
    assert d <= 1e-9 and np.array_equal(binary[13].cpu().numpy().astype(bool), rb)
