"""Pseudo-mask voting on the device (csrc/voting.hip) against the oracle (whose filter is pinned by the reference's own
utils.misc.filter_masks through tests/golden/voting.npz): surviving set, IoU table, row sums, winner."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import voting_oracle as V  # noqa: E402  (checker only)
from selfmask_amd.voting import vote_mask  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "voting.npz")
FLAGS = {"long": (True, False), "both": (True, True), "none": (False, False)}


@pytest.mark.parametrize("tag", sorted(FLAGS))
def test_vote_matches_oracle_and_reference_filter(tag):
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        masks = torch.from_numpy(g[f"masks_{i}"])
        best_mask, best, new_to_prev = vote_mask(masks.to(DEV), *FLAGS[tag])
        assert [new_to_prev[k] for k in range(len(new_to_prev))] == g[f"kept_{i}_{tag}"].tolist()  # the real filter_masks
        ref_mask, ref_best, ref_map, table, ious = V.vote_mask(masks, *FLAGS[tag])
        assert new_to_prev == ref_map and best == ref_best and torch.equal(best_mask.cpu(), ref_mask)
        last = vote_mask.last
        kept = [new_to_prev[k] for k in range(len(new_to_prev))]
        dev_table = last["iou"].cpu()[kept][:, kept]
        assert torch.equal(dev_table, table)                                   # integer counts -> identical fp32 quotients
        assert torch.allclose(last["row_sums"].cpu()[kept], ious, rtol=0, atol=2e-6)


def test_27_candidates_like_the_generator():
    """3 feature types x (2 + 3 + 4) one-hot cluster maps (mask_generator.pyc@L136-200), 27 candidates of one image."""
    rng = np.random.Generator(np.random.PCG64(7))
    h, w = 231, 317
    yy, xx = np.mgrid[:h, :w]
    cands = []
    for f in range(3):
        for k in (2, 3, 4):
            centers = rng.uniform(0.2, 0.8, size=(k, 2)) * (h, w)
            lab = np.argmin(((yy[None] - centers[:, 0, None, None]) ** 2 + (xx[None] - centers[:, 1, None, None]) ** 2), 0)
            cands += [(lab == c) for c in range(k)]
    masks = torch.from_numpy(np.stack(cands).astype(np.uint8))
    assert masks.shape[0] == 27
    best_mask, best, new_to_prev = vote_mask(masks.to(DEV))
    ref_mask, ref_best, ref_map, _, _ = V.vote_mask(masks)
    assert new_to_prev == ref_map and best == ref_best and torch.equal(best_mask.cpu(), ref_mask)


def test_everything_filtered_raises():
    with pytest.raises(ValueError, match="filtered"):
        vote_mask(torch.zeros(3, 40, 40, dtype=torch.uint8, device=DEV))
