"""The voting oracle's filter (oracle/voting_oracle.py) against what the REAL utils.misc.filter_masks kept
(tests/golden/voting.npz, oracle/gen_golden.py --only voting)."""
import os

import numpy as np
import pytest
import torch

from oracle import voting_oracle as V
from vote_known_answers import KNOWN_ANSWERS

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
    for i in range(3):  # the three blob cases (case 3 is the all-filtered set)
        best_mask, best, new_to_prev, table, ious = V.vote_mask(torch.from_numpy(g[f"masks_{i}"]))
        assert new_to_prev[best] in (0, 1, 2, 3)  # one of the blobs around the common object
        assert torch.allclose(torch.diagonal(table), torch.ones(table.shape[0]), atol=1e-6)


def test_all_filtered_returns_everything_like_the_reference():
    """golden case 3: the real filter_masks filtered every candidate and returned all six with the identity map."""
    g = np.load(GOLD)
    masks = torch.from_numpy(g["masks_3"])
    assert g["kept_3_long"].tolist() == list(range(masks.shape[0]))
    kept, new_to_prev = V.filter_masks(masks, True, False)
    assert torch.equal(kept, masks) and new_to_prev == {i: i for i in range(masks.shape[0])}


@pytest.mark.parametrize("name", sorted(KNOWN_ANSWERS))
def test_vote_known_answers(name):
    """vote_mask exists in the reference as bytecode only: hand-computed cases pin the restatement (tests/vote_known_answers.py)."""
    masks, flags, want_best, want_map, want_table = KNOWN_ANSWERS[name]()
    best_mask, best, new_to_prev, table, ious = V.vote_mask(torch.from_numpy(masks), *flags)
    assert best == want_best and new_to_prev == want_map
    assert np.array_equal(best_mask.numpy(), masks[want_map[want_best]])
    assert np.array_equal(table.numpy(), np.asarray(want_table, np.float32))


def test_cluster_oracle_against_sklearn_and_one_hot_semantics():
    """oracle/cluster_oracle.py (the checker of csrc/cluster.hip; the reference's own clusterer is absent: parity unpinned):
    Lloyd from farthest-point centres lands on scikit-learn's partition from the same centres; the one-hot / nearest step
    reproduces to_one_hot (utils/misc.py:10-35) + F.interpolate(mode="nearest")."""
    from sklearn.cluster import KMeans
    from oracle import cluster_oracle as CO
    rng = np.random.Generator(np.random.PCG64(1))
    for k in (2, 3, 4):
        c = rng.standard_normal((k, 384)).astype(np.float32) * 2
        lab = rng.integers(0, k, 500)
        x = (c[lab] + rng.standard_normal((500, 384)).astype(np.float32) * 0.35).astype(np.float32)
        got, _ = CO.kmeans(x, k, 20)
        sk = KMeans(n_clusters=k, init=x[CO.farthest_point_init(x, k)], n_init=1, algorithm="lloyd", max_iter=20, tol=0.0).fit(x)
        assert (sk.labels_ == got).mean() >= 0.999
    labels = torch.tensor([[0, 2], [1, 2]], dtype=torch.int32)
    m = CO.to_one_hot_masks(labels, 3, 2, 3, 4)
    assert m.shape == (3, 3, 4) and m.sum(0).eq(1).all()
    assert m[2].tolist() == [[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 1, 1]] and m[1].tolist() == [[0, 0, 0, 0], [0, 0, 0, 0], [1, 1, 0, 0]]
