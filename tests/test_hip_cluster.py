"""Candidate-mask extraction of the pseudo-mask generator on the device (csrc/cluster.hip, selfmask_amd.voting): the two
interpolations against PyTorch's F.interpolate, the one-hot against the reference's to_one_hot semantics, the k-means stand-in
(the reference's `clusterings` module is absent: parity UNPINNED) against its numpy restatement and scikit-learn, and the whole
chain model -> candidates -> vote."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cluster_oracle as CO  # noqa: E402  (checker only)
from oracle import voting_oracle as V  # noqa: E402
from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images  # noqa: E402
from selfmask_amd import voting as VT  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("gh,gw,scale", [(14, 14, 2), (13, 21, 2), (1, 7, 2), (5, 1, 3), (28, 28, 2)])
def test_upsample_aligned_matches_torch(gh, gw, scale):
    tok = torch.randn(2, gh * gw, 384, generator=torch.Generator().manual_seed(gh * 100 + gw))
    got = VT.upsample_tokens_aligned(tok.to(DEV), gh, gw, scale).cpu()
    ref = CO.upsample_aligned(tok, gh, gw, scale)
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2e-6


@pytest.mark.parametrize("lh,lw,k,scale,H,W", [(28, 28, 4, 8, 224, 224), (26, 42, 3, 8, 200, 333), (9, 5, 2, 4, 33, 17)])
def test_labels_to_masks_matches_one_hot_plus_nearest(lh, lw, k, scale, H, W):
    labels = torch.randint(0, k, (lh, lw), generator=torch.Generator().manual_seed(k), dtype=torch.int32)
    got = VT.labels_to_masks(labels.to(DEV), k, scale, H, W).cpu()
    assert torch.equal(got, CO.to_one_hot_masks(labels, k, scale, H, W))


def _blobs(n, k, seed, spread=0.35):
    rng = np.random.Generator(np.random.PCG64(seed))
    centres = rng.standard_normal((k, 384)).astype(np.float32) * 2
    lab = rng.integers(0, k, n)
    return (centres[lab] + rng.standard_normal((n, 384)).astype(np.float32) * spread).astype(np.float32), lab


@pytest.mark.parametrize("n,k", [(784, 2), (784, 3), (784, 4), (3136, 4), (37, 5), (4, 4)])
def test_kmeans_matches_restatement_and_sklearn(n, k):
    x, truth = _blobs(n, k, seed=n + k)
    labels, centres = VT.kmeans(torch.from_numpy(x)[None].to(DEV), k, iters=20)
    labels, centres = labels[0].cpu().numpy(), centres[0].cpu().numpy()
    ref_l, ref_c = CO.kmeans(x, k, 20)
    assert (labels == ref_l).mean() >= 0.999 and np.abs(centres - ref_c).max() <= 1e-4
    # third-party check: scikit-learn's Lloyd from the same initial centres reaches the same partition on separated blobs
    from sklearn.cluster import KMeans
    init = x[CO.farthest_point_init(x, k)]
    sk = KMeans(n_clusters=k, init=init, n_init=1, algorithm="lloyd", max_iter=20, tol=0.0).fit(x)
    assert (sk.labels_ == labels).mean() >= 0.999
    # and it is the true partition (up to a relabelling) when the blobs are separated
    if n >= 37:
        conf = np.zeros((k, k), int)
        for a, b in zip(truth, labels):
            conf[a, b] += 1
        assert (conf.max(1).sum() / n) >= 0.99
    again, _ = VT.kmeans(torch.from_numpy(x)[None].to(DEV), k, iters=20)
    assert np.array_equal(again[0].cpu().numpy(), labels)  # deterministic


def test_kmeans_is_batched_and_argument_checked():
    xs = np.stack([_blobs(100, 3, seed=s)[0] for s in (1, 2, 3)])
    lab, _ = VT.kmeans(torch.from_numpy(xs).to(DEV), 3, iters=10)
    for i in range(3):
        assert np.array_equal(lab[i].cpu().numpy(), VT.kmeans(torch.from_numpy(xs[i:i + 1]).to(DEV), 3, iters=10)[0][0].cpu().numpy())
    with pytest.raises(RuntimeError, match="sm_kmeans_f32"):
        VT.kmeans(torch.zeros(1, 3, 384, device=DEV), 5)


@pytest.mark.parametrize("patch,size", [(16, (224, 224)), (8, (120, 152)), (16, (97, 211))])
def test_extract_candidates_then_vote(patch, size):
    """model -> 9 candidates -> vote, against the oracle chain fed with the SAME device tokens (the clustering is discrete: a
    last-bit difference in the encoder output must not be allowed to flip a label in this comparison)."""
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(31, "soft", patch_size=patch), strict=True)
    m = m.to(DEV)
    H, W = size
    x = torch.from_numpy(synthetic_images(55, (1, 3, H, W))).to(DEV)
    cands = VT.extract_candidate_masks(m, x, cluster_type="kmeans")
    assert cands.shape == (9, H, W) and cands.dtype == torch.uint8
    tok = m(x, encoder_only=True)["patch_tokens"].cpu()
    gh, gw = tok.shape[1:3]
    feats = CO.upsample_aligned(tok.reshape(1, gh * gw, 384), gh, gw, 2)[0].reshape(-1, 384).numpy()
    ref = []
    for k in (2, 3, 4):
        lab, _ = CO.kmeans(feats, k, 20)
        ref.append(CO.to_one_hot_masks(torch.from_numpy(lab).reshape(2 * gh, 2 * gw), k, patch // 2, H, W))
    ref = torch.cat(ref)
    assert (cands.cpu() != ref).float().mean().item() <= 2e-3   # a few boundary pixels may fall to the other side of a tie
    for k0, k in ((0, 2), (2, 3), (5, 4)):
        assert torch.equal(cands[k0:k0 + k].sum(0).cpu(), torch.ones(H, W, dtype=torch.uint8))  # a partition of the image
    best_mask, best, new_to_prev = VT.vote_mask(cands)
    ref_mask, ref_best, ref_map, _, _ = V.vote_mask(cands.cpu())
    assert best == ref_best and new_to_prev == ref_map and torch.equal(best_mask.cpu(), ref_mask)
    # a user-supplied clusterer (the reference's class takes one): here the oracle's, through the same plumbing
    def cpu_clusterer(f, k):
        return torch.from_numpy(CO.kmeans(f[0].cpu().numpy(), k, 20)[0]).to(DEV)
    assert (VT.extract_candidate_masks(m, x, clusterer=cpu_clusterer).cpu() != ref).float().mean().item() <= 2e-3
