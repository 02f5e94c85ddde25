"""The bilateral-solver restatement (oracle/bilateral_oracle.py) against outputs of the reference's own module."""
import os

import numpy as np

from oracle import bilateral_oracle as B

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bilateral.npz")


def test_bilateral_oracle_matches_reference():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        img, tgt = g[f"img_{i}"], g[f"target_{i}"]
        soft, binary, grid = B.bilateral_solver_output(img, tgt)
        assert grid.nvertices == int(g[f"nvert_{i}"])
        assert grid.blur_nnz == list(g[f"blur_nnz_{i}"])
        d = np.abs(soft - g[f"soft_{i}"]).max()
        assert d <= 1e-9, (i, d)
        assert np.array_equal(binary, g[f"binary_{i}"])


def test_rgb2yuv_is_numpys_tensordot_bit_for_bit():
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, size=(64, 300, 3)).astype(np.uint8)
    im[0, :256, :] = np.arange(256, dtype=np.uint8)[:, None]  # exactly-grey pixels sit on bin boundaries
    ref = np.tensordot(im, B.RGB_TO_YUV, ([2], [1])) + B.YUV_OFFSET.reshape(1, 1, -1)
    assert np.array_equal(B.rgb2yuv(im), ref)
