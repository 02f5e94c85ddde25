"""The spectral-clustering restatement (oracle/cluster_oracle.py; parity UNPINNED - the reference's `clusterings` module is absent in
every form) against its third-party witness, scikit-learn's SpectralClustering / spectral_embedding on the SAME precomputed
affinity, and against known answers on separable features."""
import warnings

import numpy as np
import pytest

from oracle import cluster_oracle as CO


def blobs(n, k, seed, spread=0.35, dim=384):
    rng = np.random.Generator(np.random.PCG64(seed))
    centres = rng.standard_normal((k, dim)).astype(np.float32) * 2
    lab = np.arange(n) % k
    rng.shuffle(lab)
    return (centres[lab] + rng.standard_normal((n, dim)).astype(np.float32) * spread).astype(np.float32), lab


def scene(g, k, seed, noise=0.25, dim=384):
    """image-like features on a g x g grid: k regions (Voronoi cells of random sites) with a prototype each, blended over a few
    cells at the borders (a bilinear up-sample does that to tokens), plus noise: a CONNECTED k-NN graph with k clear clusters."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sites = rng.uniform(0.15 * g, 0.85 * g, (k, 2))
    protos = rng.standard_normal((k, dim)) * 1.5
    yy, xx = np.mgrid[0:g, 0:g]
    d = np.stack([np.hypot(yy - s[0], xx - s[1]) for s in sites], -1)  # (g, g, k)
    wgt = np.exp(-(d - d.min(-1, keepdims=True)) / 3.0)
    wgt /= wgt.sum(-1, keepdims=True)
    f = wgt @ protos + rng.standard_normal((g, g, dim)) * noise
    return f.reshape(g * g, dim).astype(np.float32), d.argmin(-1).reshape(-1)


def agreement(a, b, k):
    conf = np.zeros((k, k), int)
    for i, j in zip(a, b):
        conf[i, j] += 1
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(-conf)
    return conf[r, c].sum() / len(a)


def test_knn_indices_are_the_nearest_others_in_order():
    x, _ = blobs(60, 3, 5, dim=16)
    idx = CO.knn_indices(x, 6)
    assert idx.shape == (60, 5) and idx.dtype == np.int32
    for i in (0, 17, 59):
        d = ((x.astype(np.float64) - x[i]) ** 2).sum(1)
        d[i] = np.inf
        assert np.array_equal(idx[i], np.argsort(d, kind="stable")[:5])
    w = CO.affinity_from_knn(idx)
    assert np.array_equal(w, w.T) and set(np.unique(w)) <= {0.0, 0.5, 1.0} and np.trace(w) == 0
    # the same graph scikit-learn builds for affinity="nearest_neighbors" (include_self=True there; the self loop is dropped by the
    # Laplacian): connectivity = kneighbors_graph(n_neighbors, include_self=True); 0.5 (C + C^T)
    from sklearn.neighbors import kneighbors_graph
    c = kneighbors_graph(x.astype(np.float64), n_neighbors=6, include_self=True).toarray()
    ws = 0.5 * (c + c.T)
    np.fill_diagonal(ws, 0)
    assert np.array_equal(w, ws)


@pytest.mark.parametrize("n,k", [(300, 2), (400, 3), (784, 4)])
def test_separable_blobs_give_the_true_partition_and_agree_with_sklearn(n, k):
    x, truth = blobs(n, k, seed=n + k)
    labels, idx, vals, emb = CO.spectral_cluster(x, (k,), 10)
    assert agreement(truth, labels[k], k) == 1.0
    assert np.all(np.abs(vals) <= 1e-10)  # k connected components: eigenvalue 0 with multiplicity k
    from sklearn.cluster import SpectralClustering
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "Graph is not fully connected"
        sk = SpectralClustering(n_clusters=k, affinity="precomputed", assign_labels="kmeans", random_state=0).fit(CO.affinity_from_knn(idx))
    assert agreement(sk.labels_, labels[k], k) == 1.0


@pytest.mark.parametrize("g,k,seed", [(28, 2, 1), (28, 3, 2), (28, 4, 3), (40, 4, 4)])
def test_connected_scene_embedding_and_labels_agree_with_sklearn(g, k, seed):
    x, truth = scene(g, k, seed)
    labels, idx, vals, emb = CO.spectral_cluster(x, (2, 3, 4) if k == 4 else (k,), 10)
    w = CO.affinity_from_knn(idx)
    from scipy.sparse.csgraph import connected_components
    assert connected_components(w)[0] == 1
    assert abs(vals[0]) <= 1e-12 and vals[1] > 1e-6
    assert agreement(truth, labels[k], k) >= 0.9  # the blended borders belong to either side
    # witness 1: scikit-learn's embedding of the same affinity spans the same subspace (columns up to sign / rotation)
    from sklearn.manifold import spectral_embedding
    se = spectral_embedding(w, n_components=k, eigen_solver="arpack", random_state=0, drop_first=False, eigen_tol=1e-12)
    q1, _ = np.linalg.qr(emb[:, :k])
    q2, _ = np.linalg.qr(se)
    assert np.linalg.svd(q1.T @ q2, compute_uv=False).min() >= 1 - 1e-8
    # witness 2: its SpectralClustering (random k-means++ restarts instead of farthest-point centres) finds the same partition
    from sklearn.cluster import SpectralClustering
    sk = SpectralClustering(n_clusters=k, affinity="precomputed", assign_labels="kmeans", random_state=0).fit(w)
    assert agreement(sk.labels_, labels[k], k) >= 0.99
    # the eigenpairs are eigenpairs
    d = w.sum(1)
    lap = np.eye(len(d)) - w / np.sqrt(d)[:, None] / np.sqrt(d)[None, :]
    v = emb * np.sqrt(d)[:, None]
    assert np.abs(lap @ v - v * vals[None]).max() <= 1e-12


def test_kmeans_embedding_is_invariant_to_sign_and_rotation_of_the_columns():
    x, _ = scene(28, 3, 9)
    _, idx, vals, emb = CO.spectral_cluster(x, (3,), 10)
    a = CO.kmeans_embedding(emb[:, :3], 3)
    rng = np.random.Generator(np.random.PCG64(0))
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    assert np.array_equal(a, CO.kmeans_embedding(emb[:, :3] @ q, 3))
    assert np.array_equal(a, CO.kmeans_embedding(emb[:, :3] * np.array([1, -1, -1.0]), 3))
