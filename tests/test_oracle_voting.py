"""The voting oracle's filter (oracle/voting_oracle.py) against what the REAL utils.misc.filter_masks kept
(tests/golden/voting.npz, oracle/gen_golden.py --only voting)."""
import os

import numpy as np
import pytest
import torch

from oracle import voting_oracle as V

GOLD = os.path.join(os.path.dirname(__file__), "golden", "voting.npz")
FLAGS = {"long": (True, False), "both": (True, True), "none": (False, False)}


@pytest.mark.parametrize("tag", sorted(FLAGS))
def test_filter_matches_reference(tag):
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        masks = torch.from_numpy(g[f"masks_{i}"])
        kept, new_to_prev = V.filter_masks(masks, *FLAGS[tag])
        assert [new_to_prev[k] for k in range(len(new_to_prev))] == g[f"kept_{i}_{tag}"].tolist()
        assert kept.shape[0] == len(new_to_prev)


def test_vote_picks_the_consensus_mask():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        best_mask, best, new_to_prev, table, ious = V.vote_mask(torch.from_numpy(g[f"masks_{i}"]))
        assert new_to_prev[best] in (0, 1, 2, 3)  # one of the blobs around the common object
        assert torch.allclose(torch.diagonal(table), torch.ones(table.shape[0]), atol=1e-6)
