"""ORACLE (test infrastructure, NOT product code): CPU restatement of the candidate-mask extraction of the pseudo-mask
generator, DINO branch (datasets/mask_generator, bytecode only; SURVEY.md Appendix B @L136-200).

Parity status: the two interpolations and ``to_one_hot`` are PyTorch's / the reference's own (``F.interpolate`` calls as the
bytecode makes them; ``to_one_hot`` follows /root/reference/utils/misc.py:10-35).  The CLUSTERING is UNPINNED: the reference's
``clusterings`` module (KMeansClustering / SpectralClustering) is absent from its repository in every form, so there is nothing to
pin against - ``kmeans`` restates the product's stand-in (csrc/cluster.hip: Lloyd iterations from farthest-point initial centres,
sums in a fixed order) in numpy float32, and tests/test_hip_cluster.py additionally runs scikit-learn's KMeans from the same
initial centres as a third-party check.

``cluster_type="spectral"`` (the shipped YAML's choice, /root/reference/configs/duts-dino-k234-nq20-224-swav-mocov2-dino-p16-
sr10100.yaml:11-12; BASELINE configs[4]: "faiss k-NN affinity + eigendecomp") is UNPINNED for the same reason.  What is restated
below is the textbook algorithm the paper names (arXiv 2203.12614 section 3.1 "spectral clustering" over the self-supervised
features; Shi & Malik normalised cuts / von Luxburg's tutorial, algorithm "normalized spectral clustering"), in exactly the form
scikit-learn's ``SpectralClustering(affinity="precomputed")`` evaluates it, so that scikit-learn can serve as the third-party
witness (tests/test_oracle_spectral.py):
    k-NN graph (Euclidean, every point + its n_neighbors - 1 nearest others = faiss IndexFlatL2.search(x, n_neighbors), which
    returns the query itself first) -> connectivity C -> W = (C + C^T) / 2 with the self loops dropped -> d = W 1 ->
    L = I - D^-1/2 W D^-1/2 -> eigenvectors v_1..v_k of the k smallest eigenvalues -> embedding rows u_i = v_i / sqrt(d_i)
    (sklearn.manifold.spectral_embedding(norm_laplacian=True, drop_first=False)) -> k-means on the rows.
The k-means on the embedding is the same deterministic Lloyd iteration as above (farthest-point initial centres), in float64."""
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


# ---------------------------------------------------------------------------------------------------------------------------
# spectral clustering (parity UNPINNED: see the header)


def knn_indices(x: np.ndarray, n_neighbors: int) -> np.ndarray:
    """x (n, D) -> (n, n_neighbors - 1) int32: for every point its nearest OTHER points by (squared Euclidean distance, index),
    ascending.  With the point itself in front this is what a flat L2 k-NN index returns for a query that is in the index."""
    x = x.astype(np.float64)
    n = x.shape[0]
    m = min(n_neighbors - 1, n - 1)
    sq = (x * x).sum(1)
    d = sq[:, None] + sq[None, :] - 2.0 * (x @ x.T)
    d[np.arange(n), np.arange(n)] = -np.inf  # the point itself sorts first and is dropped
    return np.argsort(d, axis=1, kind="stable")[:, 1:m + 1].astype(np.int32)


def knn_boundary_gap(x: np.ndarray, n_neighbors: int) -> np.ndarray:
    """(n,) the distance gap between the last neighbour kept and the first one left out: rows with a tiny gap may legitimately
    pick the other one on a kernel whose products carry 22 bits."""
    x = x.astype(np.float64)
    n = x.shape[0]
    m = min(n_neighbors - 1, n - 1)
    sq = (x * x).sum(1)
    d = sq[:, None] + sq[None, :] - 2.0 * (x @ x.T)
    d[np.arange(n), np.arange(n)] = -np.inf
    s = np.sort(d, axis=1)
    return (s[:, m + 1] - s[:, m]) if m + 1 < n else np.full(n, np.inf)


def affinity_from_knn(idx: np.ndarray) -> np.ndarray:
    """(n, m) neighbour lists -> dense W (n, n) float64: (C + C^T) / 2, zero diagonal; entries in {0, 0.5, 1}."""
    n = idx.shape[0]
    c = np.zeros((n, n))
    c[np.repeat(np.arange(n), idx.shape[1]), idx.reshape(-1)] = 1.0
    w = 0.5 * (c + c.T)
    w[np.arange(n), np.arange(n)] = 0.0
    return w


def spectral_embedding(w: np.ndarray, kw: int):
    """dense affinity -> (eigenvalues (kw,) of L = I - D^-1/2 W D^-1/2 ascending, eigenvectors V (n, kw), embedding V / sqrt(d))."""
    from scipy.linalg import eigh
    d = w.sum(1)
    dd = np.sqrt(d)
    lap = np.eye(w.shape[0]) - w / dd[:, None] / dd[None, :]
    vals, vecs = eigh(lap, subset_by_index=[0, kw - 1])
    return vals, vecs, vecs / dd[:, None]


def kmeans_embedding(e: np.ndarray, k: int, max_iter: int = 100):
    """rows of e (n, k) float64 -> labels: farthest-point initial centres (first the point farthest from the mean), Lloyd until
    no label changes (at most max_iter updates); ties to the lowest index, an emptied cluster keeps its centre."""
    e = e.astype(np.float64)
    ref = e.mean(0)
    picks, mind = [], None
    for j in range(k):
        d = ((e - ref) ** 2).sum(1)
        if j > 0:
            mind = d if mind is None else np.minimum(mind, d)
            d = mind
        picks.append(int(np.argmax(d)))
        ref = e[picks[-1]]
    cen = e[picks].copy()
    labels = None
    for it in range(max_iter + 1):
        new = ((e[:, None, :] - cen[None]) ** 2).sum(-1).argmin(1)
        if labels is not None and np.array_equal(new, labels):
            break
        labels = new
        if it == max_iter:
            break
        for c in range(k):
            msk = labels == c
            if msk.any():
                cen[c] = e[msk].mean(0)
    return labels.astype(np.int32)


def spectral_cluster(x: np.ndarray, cluster_sizes=(2, 3, 4), n_neighbors: int = 10):
    """x (n, D) float32 -> {k: labels (n,)}: ONE eigen-decomposition for max(cluster_sizes) vectors, the first k of them per k."""
    idx = knn_indices(x, n_neighbors)
    vals, vecs, emb = spectral_embedding(affinity_from_knn(idx), max(cluster_sizes))
    return {k: kmeans_embedding(emb[:, :k], k) for k in cluster_sizes}, idx, vals, emb
