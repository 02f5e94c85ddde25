"""ORACLE (test infrastructure, NOT product code): CPU restatement of the candidate-mask extraction of the pseudo-mask
generator, DINO branch (datasets/mask_generator, bytecode only; SURVEY.md Appendix B @L136-200).

Parity status: the two interpolations and ``to_one_hot`` are PyTorch's / the reference's own (``F.interpolate`` calls as the
bytecode makes them; ``to_one_hot`` follows /root/reference/utils/misc.py:10-35).  The CLUSTERING is UNPINNED: the reference's
``clusterings`` module (KMeansClustering / SpectralClustering) is absent from its repository in every form, so there is nothing to
pin against - ``kmeans`` restates the product's stand-in (csrc/cluster.hip: Lloyd iterations from farthest-point initial centres,
sums in a fixed order) in numpy float32, and tests/test_hip_cluster.py additionally runs scikit-learn's KMeans from the same
initial centres as a third-party check."""
import numpy as np
import torch
import torch.nn.functional as F


def upsample_aligned(tokens: torch.Tensor, gh: int, gw: int, scale: int = 2) -> torch.Tensor:
    """mask_generator.pyc@L159: tokens (B, gh*gw, D) -> (B, s gh, s gw, D) via F.interpolate(..., align_corners=True)."""
    B, n, D = tokens.shape
    f = tokens.reshape(B, gh, gw, D).permute(0, 3, 1, 2)
    return F.interpolate(f, scale_factor=scale, mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()


def to_one_hot_masks(labels: torch.Tensor, k: int, scale: int, H: int, W: int) -> torch.Tensor:
    """utils/misc.py:10-35 (k given) + mask_generator.pyc@L161: F.interpolate(one_hot[None], scale_factor, mode="nearest")[0][..., :H, :W]."""
    h, w = labels.shape
    one_hot = torch.zeros((h * w, k), dtype=torch.long)
    one_hot.scatter_(1, labels.reshape(-1, 1).long(), torch.ones((h * w, 1), dtype=torch.long))
    one_hot = one_hot.view(h, w, k).permute(2, 0, 1)
    up = F.interpolate(one_hot[None].float(), scale_factor=(scale, scale), mode="nearest")[0][..., :H, :W]
    return up.to(torch.uint8)


def farthest_point_init(x: np.ndarray, k: int) -> np.ndarray:
    """indices of the initial centres: the point farthest from the mean, then repeatedly the point farthest from its nearest
    chosen centre (first maximum)."""
    x = x.astype(np.float32)
    ref = x.mean(0, dtype=np.float32)
    picks, mind = [], None
    for j in range(k):
        d = ((x - ref) ** 2).sum(1, dtype=np.float32)
        if j > 0:
            mind = d if mind is None else np.minimum(mind, d)
            d = mind
        i = int(np.argmax(d))
        picks.append(i)
        ref = x[i]
    return np.array(picks)


def kmeans(x: np.ndarray, k: int, iters: int = 20):
    """x (n, D) -> (labels (n,), centres (k, D)); ties to the lowest cluster index, an emptied cluster keeps its centre."""
    x = x.astype(np.float32)
    cen = x[farthest_point_init(x, k)].copy()
    for it in range(iters + 1):
        d = ((x[:, None, :] - cen[None]) ** 2).sum(-1, dtype=np.float32)
        labels = d.argmin(1)
        if it == iters:
            break
        for c in range(k):
            m = labels == c
            if m.any():
                cen[c] = x[m].sum(0, dtype=np.float32) / np.float32(m.sum())
    return labels.astype(np.int32), cen
