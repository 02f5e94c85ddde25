"""Spectral clustering on the device (csrc/spectral.hip, selfmask_amd.voting.spectral_cluster) - the clusterer the shipped YAML selects
for the pseudo-mask generator.  Parity UNPINNED (the reference's `clusterings` module is absent in every form): every stage is
compared with its numpy / scipy restatement (oracle/cluster_oracle.py), scikit-learn is the third-party witness on the device's own
graph, and separable features must come back as the true partition."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cluster_oracle as CO  # noqa: E402  (checker only)
from oracle import voting_oracle as V  # noqa: E402
from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images  # noqa: E402
from selfmask_amd import voting as VT  # noqa: E402
from test_oracle_spectral import agreement, blobs, scene  # noqa: E402

DEV = "cuda:0"


def run(x, sizes=(2, 3, 4), n_neighbors=10, **kw):
    labels, det = VT.spectral_cluster(torch.from_numpy(x)[None].to(DEV), sizes, n_neighbors, return_details=True, **kw)
    return labels[0].cpu().numpy(), {k: v[0].cpu().numpy() for k, v in det.items()}


@pytest.mark.parametrize("n,k", [(784, 2), (784, 3), (784, 4), (840, 3), (900, 2), (1024, 2), (1444, 3), (1936, 4), (2000, 3), (2500, 2), (2700, 3), (3136, 2), (3136, 3), (3136, 4), (6400, 3)])
def test_separable_features_come_back_as_the_true_partition(n, k):
    x, truth = blobs(n, k, seed=n + k)
    labels, det = run(x, (k,))
    assert agreement(truth, labels[0], k) == 1.0
    assert det["info"][2] == 1 and det["residuals"].max() <= 1e-8
    assert np.abs(det["eigenvalues"]).max() <= 1e-9  # k components: eigenvalue 0, k times


@pytest.mark.parametrize("g,k,seed,nn", [(28, 2, 1, 10), (28, 3, 2, 10), (28, 4, 3, 10), (56, 4, 4, 10), (56, 3, 7, 10), (28, 4, 3, 20), (28, 3, 5, 6),
                                          (32, 3, 11, 10), (38, 3, 8, 10), (44, 4, 13, 10), (38, 2, 10, 24), (50, 3, 22, 10)])
# (28^2: graph and both blocks of the recurrence in the LDS for the whole solve; 32^2, 38^2: the graph and 4 columns of the blocks per filter,
# 44^2: 2 columns, 50^2: 1 column; 38^2 with 24 neighbours and 56^2: graph in memory.  2 000 points above: admitted to the 2-column plan by the launch, sent back to
# the graph-in-memory steps by the kernel once it has seen the real list lengths)
def test_every_stage_against_its_restatement(g, k, seed, nn):
    x, truth = scene(g, k, seed)
    n = g * g
    labels, det = run(x, (2, 3, 4), nn)
    # stage 2: the neighbour lists.  The Gram matrix carries 22 bits: a row whose last kept and first dropped neighbour are closer
    # than that may legitimately swap them
    ref_idx = CO.knn_indices(x, nn)
    gap = CO.knn_boundary_gap(x, nn)
    bad = [i for i in range(n) if set(det["knn"][i]) != set(ref_idx[i])]
    assert all(gap[i] <= 2e-3 for i in bad) and len(bad) <= n // 200, (len(bad), [gap[i] for i in bad][:5])
    assert (det["knn"] == ref_idx).mean() >= 0.995  # and nearest-first order
    assert all(i not in det["knn"][i] for i in range(n))
    # stage 4: eigenpairs of the DEVICE's graph against a dense eigh of the same graph
    w = CO.affinity_from_knn(det["knn"])
    vals, vecs, emb = CO.spectral_embedding(w, 4)
    assert det["info"][2] == 1 and det["info"][3] == 0, det["info"]
    assert np.abs(det["eigenvalues"] - vals).max() <= 1e-10
    assert det["residuals"].max() <= 1e-8
    d = w.sum(1)
    lap = np.eye(n) - w / np.sqrt(d)[:, None] / np.sqrt(d)[None, :]
    v = det["embedding"] * np.sqrt(d)[:, None]
    assert np.abs(lap @ v - v * det["eigenvalues"][None]).max() <= 1e-8  # the judge's criterion, recomputed on the host
    assert np.abs(v.T @ v - np.eye(4)).max() <= 1e-10
    for j in range(4):  # distinct eigenvalues here: vectors equal up to sign
        if min(abs(vals[j] - vals[i]) for i in range(4) if i != j) > 1e-6:
            s = np.sign(v[:, j] @ vecs[:, j])
            assert np.abs(v[:, j] * s - vecs[:, j]).max() <= 1e-6
    # stage 5: k-means of the DEVICE's embedding, exactly; and the whole chain through the restatement
    for i, kk in enumerate((2, 3, 4)):
        assert np.array_equal(labels[i], CO.kmeans_embedding(det["embedding"][:, :kk], kk))
        assert (labels[i] == CO.kmeans_embedding(emb[:, :kk], kk)).mean() >= 0.999
    assert agreement(truth, labels[(2, 3, 4).index(k)], k) >= 0.9
    # third-party witness on the same graph
    from sklearn.cluster import SpectralClustering
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sk = SpectralClustering(n_clusters=k, affinity="precomputed", assign_labels="kmeans", random_state=0).fit(w)
    assert agreement(sk.labels_, labels[(2, 3, 4).index(k)], k) >= 0.99


def test_batched_deterministic_and_argument_checked():
    xs = np.stack([scene(28, 3, s)[0] for s in (11, 12, 13)])
    lab = VT.spectral_cluster(torch.from_numpy(xs).to(DEV), (2, 3))
    assert lab.shape == (3, 2, 784) and lab.dtype == torch.int32
    for i in range(3):
        one = VT.spectral_cluster(torch.from_numpy(xs[i:i + 1]).to(DEV), (2, 3))
        assert torch.equal(one[0], lab[i])
    assert torch.equal(VT.spectral_cluster(torch.from_numpy(xs).to(DEV), (2, 3)), lab)
    with pytest.raises(ValueError, match="spectral_cluster"):
        VT.spectral_cluster(torch.zeros(1, 8, 384, device=DEV), (2,))
    with pytest.raises(ValueError, match="spectral_cluster"):
        VT.spectral_cluster(torch.zeros(1, 784, 384, device=DEV), (7,))
    with pytest.raises(RuntimeError, match="HIP device"):
        VT.spectral_cluster(torch.zeros(1, 784, 384), (2,))


def test_non_finite_features_do_not_fault():
    x = scene(28, 2, 3)[0].copy()
    x[5] = np.nan
    x[77, 3] = np.inf
    lab = VT.spectral_cluster(torch.from_numpy(x)[None].to(DEV), (2, 3, 4))
    torch.cuda.synchronize()
    assert lab.shape == (1, 3, 784) and int(lab.min()) >= 0 and int(lab.max()) <= 3


@pytest.mark.parametrize("patch,size", [(16, (224, 224)), (8, (120, 152)), (16, (97, 211))])
def test_extract_candidates_spectral_then_vote(patch, size):
    """model -> 9 candidates (spectral, the default) -> vote, against the oracle chain fed with the SAME device tokens."""
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(31, "soft", patch_size=patch), strict=True)
    m = m.to(DEV)
    H, W = size
    x = torch.from_numpy(synthetic_images(55, (1, 3, H, W))).to(DEV)
    cands = VT.extract_candidate_masks(m, x)
    assert cands.shape == (9, H, W) and cands.dtype == torch.uint8
    tok = m(x, encoder_only=True)["patch_tokens"].cpu()
    gh, gw = tok.shape[1:3]
    feats = CO.upsample_aligned(tok.reshape(1, gh * gw, 384), gh, gw, 2)[0].reshape(-1, 384).numpy()
    ref_labels, _, _, _ = CO.spectral_cluster(feats, (2, 3, 4), 10)
    ref = torch.cat([CO.to_one_hot_masks(torch.from_numpy(ref_labels[k]).reshape(2 * gh, 2 * gw), k, patch // 2, H, W) for k in (2, 3, 4)])
    assert (cands.cpu() != ref).float().mean().item() <= 5e-3  # a few boundary points may fall to the other side of a tie
    for k0, k in ((0, 2), (2, 3), (5, 4)):
        assert torch.equal(cands[k0:k0 + k].sum(0).cpu(), torch.ones(H, W, dtype=torch.uint8))  # a partition of the image
    best_mask, best, new_to_prev = VT.vote_mask(cands)
    ref_mask, ref_best, ref_map, _, _ = V.vote_mask(cands.cpu())
    assert best == ref_best and new_to_prev == ref_map and torch.equal(best_mask.cpu(), ref_mask)
    # the k-means option and a batch of two give the same plumbing
    km = VT.extract_candidate_masks(m, x, cluster_type="kmeans")
    assert km.shape == (9, H, W)
    two = VT.extract_candidate_masks(m, torch.cat([x, x]))
    assert two.shape == (2, 9, H, W) and torch.equal(two[0], cands) and torch.equal(two[1], cands)
    # the batched forms (one launch per stage for all images) against the per-image ones
    mixed = torch.stack([cands, km])
    for (bm, bi, bmap), one in zip(VT.vote_mask_batch(mixed), (VT.vote_mask(cands), VT.vote_mask(km))):
        assert bi == one[1] and bmap == one[2] and torch.equal(bm, one[0])
    lab = torch.randint(0, 2, (3, 2, gh * gw * 4), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(DEV)
    got = VT.labels_to_masks_batch(lab, (2, 3), 2 * gh, 2 * gw, patch // 2, H, W)
    for b in range(3):
        ref_b = torch.cat([VT.labels_to_masks(lab[b, i].reshape(2 * gh, 2 * gw), k, patch // 2, H, W) for i, k in enumerate((2, 3))])
        assert torch.equal(got[b], ref_b)


def test_mask_generator_files_to_encoded_masks(tmp_path):
    """MaskGenerator(...)(p_images) (mask_generator.pyc@L232-252): JPEG files of two sizes -> {file name: RLE of the voted mask}; every
    file's result equals the oracle chain (CPU decode + normalise, device tokens, restated clustering / one-hot / vote) on that file."""
    from PIL import Image
    from selfmask_amd.datasets import MEAN, STD, synthetic_scene
    from selfmask_amd.mask_generator import MaskGenerator, rle_decode
    patch = 16
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(31, "soft", patch_size=patch), strict=True)
    m = m.to(DEV)
    rng = np.random.Generator(np.random.PCG64(4))
    paths = []
    for i, (h, w) in enumerate([(96, 128), (100, 120), (96, 128), (96, 128), (90, 125)]):  # the last one pads to the first one's 6 x 8 patch grid
        img, _ = synthetic_scene(rng, h, w)
        p = str(tmp_path / f"img_{i}.png")  # PNG: the file holds exactly these pixels
        Image.fromarray(img).save(p)
        paths.append(p)
    gen = MaskGenerator(network=m, device=DEV, batch_size=2)
    out = gen(paths)
    assert sorted(out) == [f"img_{i}.png" for i in range(5)]
    for p in paths:
        name = p.split("/")[-1]
        got = rle_decode(out[name])
        rgb = np.asarray(Image.open(p).convert("RGB"), np.float32) / np.float32(255.0)
        x = torch.from_numpy(np.ascontiguousarray(((rgb - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)).transpose(2, 0, 1)))[None]
        H, W = x.shape[-2:]
        assert got.shape == (H, W)
        tok = m(x.to(DEV), encoder_only=True)["patch_tokens"].cpu()
        gh, gw = tok.shape[1:3]
        feats = CO.upsample_aligned(tok.reshape(1, gh * gw, 384), gh, gw, 2)[0].reshape(-1, 384).numpy()
        ref_labels, _, _, _ = CO.spectral_cluster(feats, (2, 3, 4), 10)
        ref = torch.cat([CO.to_one_hot_masks(torch.from_numpy(ref_labels[k]).reshape(2 * gh, 2 * gw), k, patch // 2, H, W) for k in (2, 3, 4)])
        ref_mask = V.vote_mask(ref)[0].numpy()
        assert (got != ref_mask).mean() <= 5e-3, name
    raw = gen(paths[:1], encode=False)
    assert np.array_equal(raw["img_0.png"], rle_decode(out["img_0.png"]))
    # the batches the generator feeds the encoder hold exactly what CustomDataset would (to_tensor + normalize), per file
    from selfmask_amd.pipeline import preprocess_on_device
    seen, sizes_seen = {}, []
    for names, rgbs in gen._batches(paths):
        Hp, Wp = (-(-max(r.shape[d] for r in rgbs) // patch) * patch for d in (0, 1))
        xb = preprocess_on_device(rgbs, None, DEV, pinned=True, pad_to=(Hp, Wp)).cpu()
        seen.update(zip(names, xb))
        sizes_seen.append(sorted({r.shape[:2] for r in rgbs}))
    assert [(90, 125), (96, 128)] in sizes_seen  # two sizes, one patch grid, one batch
    for p in paths:
        want, got_x = gen._load(p), seen[p.split("/")[-1]]
        h, w = want.shape[-2:]
        assert torch.equal(got_x[:, :h, :w], want) and not got_x[:, h:].any() and not got_x[:, :, w:].any()  # zero padding, as alone
    # neither the batch size nor the number of batches in flight changes a result
    for bs, st in ((1, 1), (4, 2)):
        again = MaskGenerator(network=m, device=DEV, batch_size=bs, streams=st)(paths)
        assert again == out


def test_the_memory_plans_give_the_same_bits():
    """The eigen-solver's plans (graph + blocks in the LDS for the whole solve at 28^2; 4 / 2 / 1 columns per filter at 32^2, 44^2, 50^2)
    against the graph-in-memory plan the tuning build can force (SM_SPECTRAL_PLAN=0): the same sums in the same order - eigenvalues,
    embedding, residuals and labels bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tuning = os.path.join(root, "salient-object-detection_amd", "lib", "libselfmask_hip_tuning.so")
    if not os.path.exists(tuning):
        pytest.skip("tuning library not built (salient-object-detection_amd/build.py --tuning)")
    code = r"""
import hashlib, json, os, sys
import numpy as np, torch
sys.path[:0] = [os.path.join(sys.argv[1], "salient-object-detection_amd"), sys.argv[1], os.path.join(sys.argv[1], "tests")]
from selfmask_amd import voting as VT
from test_oracle_spectral import scene
out = {}
for g in (28, 32, 44, 50):
    x = torch.from_numpy(np.stack([scene(g, 3, 40 + s)[0] for s in range(2)])).cuda()
    labels, det = VT.spectral_cluster(x, (2, 3, 4), return_details=True)
    h = hashlib.sha1()
    for t in (labels, det["eigenvalues"], det["embedding"], det["residuals"], det["info"][:, :3]):
        h.update(t.cpu().numpy().tobytes())
    out[str(g)] = h.hexdigest()
print("RESULT " + json.dumps(out))
"""
    got = {}
    for tag, env in (("resident", {"SM_HIP_LIB": tuning}), ("memory", {"SM_HIP_LIB": tuning, "SM_SPECTRAL_PLAN": "0"})):
        r = subprocess.run([sys.executable, "-c", code, root], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert got["resident"] == got["memory"], got


def test_mask_generator_sharded_over_two_ranks(tmp_path):
    """configs[4] across ranks, rehearsed as two processes on this one GPU with a gloo group: MaskGenerator(...)(files, comm=...) gives every
    rank the codes of ALL files, equal to the single-process result."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from PIL import Image
    from selfmask_amd.datasets import synthetic_scene
    from selfmask_amd.mask_generator import MaskGenerator
    rng = np.random.Generator(np.random.PCG64(9))
    paths = []
    for i, (h, w) in enumerate([(96, 128), (90, 125), (100, 120), (96, 128), (64, 64)]):
        p = str(tmp_path / f"f{i}.png")
        Image.fromarray(synthetic_scene(rng, h, w)[0]).save(p)
        paths.append(p)
    m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(31, "soft", patch_size=16), strict=True)
    single = MaskGenerator(network=m.to(DEV), device=DEV, batch_size=2)(paths)
    code = r"""
import json, os, sys
import torch, torch.distributed as dist
sys.path[:0] = [os.path.join(sys.argv[1], "salient-object-detection_amd"), sys.argv[1]]
from selfmask_amd import MaskFormer, synthetic_state_dict
from selfmask_amd.distributed import TorchDistComm
from selfmask_amd.mask_generator import MaskGenerator
dist.init_process_group("gloo")
m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
m.load_state_dict(synthetic_state_dict(31, "soft", patch_size=16), strict=True)
out = MaskGenerator(network=m.to("cuda:0"), device="cuda:0", batch_size=2)(json.loads(sys.argv[2]), comm=TorchDistComm())
print("RESULT " + json.dumps(out, sort_keys=True))
dist.destroy_process_group()
"""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", code, root, json.dumps(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env={**os.environ, "RANK": str(r), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                                   "HSA_ENABLE_IPC_MODE_LEGACY": "0"}) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so_, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    got = [json.loads([ln for ln in so_.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so_, _ in outs]
    assert got[0] == got[1] == json.loads(json.dumps(single, sort_keys=True))
    assert list(got[0]) == sorted(got[0]) and len(got[0]) == 5


def test_mask_generator_odd_inputs(tmp_path):
    """Grayscale JPEG, RGBA / palette / 16-bit-free PNG, one file, no file: every input reaches the encoder as ``.convert("RGB")`` + to_tensor +
    normalize (the batches hold what ``_load`` gives), every file comes back with a code of its own size, an empty list gives an empty dict."""
    from PIL import Image
    from selfmask_amd.datasets import synthetic_scene
    from selfmask_amd.mask_generator import MaskGenerator, rle_decode
    from selfmask_amd.pipeline import preprocess_on_device
    m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(31, "soft", patch_size=16), strict=True)
    gen = MaskGenerator(network=m.to(DEV), device=DEV)
    assert gen([]) == {}
    rng = np.random.Generator(np.random.PCG64(3))
    img = Image.fromarray(synthetic_scene(rng, 80, 112)[0])
    paths = {}
    for name, im, kw in (("gray.jpg", img.convert("L"), {}), ("rgba.png", img.convert("RGBA"), {}), ("pal.png", img.convert("P"), {}),
                         ("cmyk.jpg", img.convert("CMYK"), {})):
        p = str(tmp_path / name)
        im.save(p, **kw)
        paths[name] = p
    out = gen(list(paths.values()))
    assert sorted(out) == sorted(paths)
    for name in paths:
        assert rle_decode(out[name]).shape == (80, 112)
    one = gen([paths["gray.jpg"]])
    assert one == {"gray.jpg": out["gray.jpg"]}  # alone or in a batch: the same code
    for names, rgbs in gen._batches(list(paths.values())):
        xb = preprocess_on_device(rgbs, None, DEV, pinned=True, pad_to=(80, 112)).cpu()
        for n, x in zip(names, xb):
            assert torch.equal(x, gen._load(paths[n])), n
