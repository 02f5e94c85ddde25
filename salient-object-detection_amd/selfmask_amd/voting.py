"""Mirror of the reference's pseudo-mask generator, DINO branch (datasets/mask_generator, bytecode only: SURVEY.md Appendix B):
``extract_candidate_masks`` (@L136-200: encoder tokens -> bilinear x2 ``align_corners=True`` -> ``clusterer(features, k)`` for
k in {2, 3, 4} -> one-hot -> nearest up-sample to the image) and ``vote_mask`` (@L202-230) with ``utils.misc.filter_masks``
(utils/misc.py:285-314), all on the MI355X.

The reference's ``clusterings`` module (``KMeansClustering`` / ``SpectralClustering``: faiss k-NN affinity + eigen-decomposition)
does not exist in its repository in any form, so BOTH clusterers here are parity UNPINNED: ``spectral_cluster`` (the default, as
in the shipped YAML: ``clustering_mode: "spectral"``, configs/duts-dino-k234-nq20-224-swav-mocov2-dino-p16-sr10100.yaml:11-12;
csrc/spectral.hip: k-NN graph -> normalised Laplacian -> eigenvectors -> k-means, the algorithm scikit-learn's
``SpectralClustering(affinity="precomputed")`` evaluates, which is its third-party witness) and ``kmeans`` (the
``cluster_type="kmeans"`` option, csrc/cluster.hip); any other clusterer can be passed as a callable, as the reference's class
takes one.  The ResNet-50 MoCo-v2 / SwAV feature branches are out of scope."""
import ctypes as C
from typing import Dict, Sequence, Tuple

import torch

from . import _native as N

# MaskGenerator.__init__ defaults and CLI choices (mask_generator.pyc@L21-38,255-294), pinned by tests/golden/evaluator_constants.json
CLUSTER_TYPES = ("k-means", "spectral")
DEFAULT_CLUSTER_SIZES = (2, 3, 4)
DEFAULT_CLUSTER_TYPE = "spectral"
FEATURE_UPSAMPLE = {"scale_factor": 2, "mode": "bilinear", "align_corners": True}  # mask_generator.pyc@L159
MASK_UPSAMPLE_MODE = "nearest"                                                       # mask_generator.pyc@L161
DINO_TOTAL_STRIDE = 16                                                               # mask_generator.pyc@L150-157 (ResNets: 8)


def vote_mask(batch_pred_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False):
    """batch_pred_masks (M, H, W) 0/1 on a HIP device -> (best mask (H, W), best index among the SURVIVORS,
    new_index_to_prev_index) as the reference returns them, plus the device tensors under ``vote_mask.last`` for inspection.
    When every candidate is filtered the reference catches the empty ``torch.stack`` and votes over ALL candidates with the
    identity index map (utils/misc.py:311-314); the kernel does the same (``keep`` comes back all ones)."""
    if not batch_pred_masks.is_cuda:
        raise RuntimeError("vote_mask (MI355X) needs its candidates on a HIP device; there is no CPU fallback")
    m = batch_pred_masks.to(torch.uint8).contiguous()
    M, H, W = m.shape
    lib = N.load()
    dev = m.device
    nbytes = lib.sm_vote_workspace_bytes(M, H, W)
    if nbytes == 0:
        raise ValueError(f"vote_mask: {M} candidates of {H}x{W} (1..64 candidates)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    keep = torch.empty(M, dtype=torch.int32, device=dev)
    iou = torch.empty((M, M), dtype=torch.float32, device=dev)
    sums = torch.empty(M, dtype=torch.float32, device=dev)
    best = torch.empty(1, dtype=torch.int32, device=dev)
    N.check(lib.sm_vote_masks_u8(m.data_ptr(), M, H, W, int(remove_long_masks), int(remove_small_large_masks), keep.data_ptr(),
                                 iou.data_ptr(), sums.data_ptr(), best.data_ptr(), ws.data_ptr(), nbytes,
                                 torch.cuda.current_stream(dev).cuda_stream), "sm_vote_masks_u8")
    keep_h, best_h = keep.cpu().tolist(), int(best.cpu()[0])
    vote_mask.last = {"keep": keep, "iou": iou, "row_sums": sums, "best": best}
    if best_h < 0:  # unreachable since the all-filtered fallback (kept as a guard against a broken launch)
        raise RuntimeError("sm_vote_masks_u8 returned no winner")
    new_to_prev: Dict[int, int] = {}
    for prev, k in enumerate(keep_h):
        if k:
            new_to_prev[len(new_to_prev)] = prev
    prev_to_new = {v: k for k, v in new_to_prev.items()}
    return batch_pred_masks[best_h], prev_to_new[best_h], new_to_prev


_ARENAS = {}


def _arena(tag: str, nbytes: int, dev) -> torch.Tensor:
    """``nbytes`` of scratch on ``dev`` for the launches a call queues on the CURRENT stream: one grow-only buffer per (tag, device,
    stream).  Launches on one stream run in order, so the next call's scratch may be the same memory; nothing handed back to a caller
    lives here.  (Fresh torch.empty buffers per call - a Gram matrix is hundreds of MB and its size changes with every patch grid -
    had the caching allocator free and re-allocate device memory batch after batch: a third of the host time of a mixed-size run.)"""
    dev = torch.device(dev)
    key = (tag, dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _ARENAS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _ARENAS.pop(key, None)
        buf = _ARENAS[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
    return buf[:nbytes]


def release_scratch() -> None:
    """Drop every per-stream scratch buffer of this module (they are grow-only: a run over very large batches keeps its high-water mark)."""
    _ARENAS.clear()


class PendingVotes:
    """The votes of one batch, queued on a stream: ``result()`` waits for THAT batch's device-to-host copy only, so the caller can
    queue the next batch (on another stream) before asking.  ``winners`` (B, H, W) uint8 is the voted mask of every image on the
    device; ``winners_host()`` the same in page-locked host memory (one copy for the batch)."""

    def __init__(self, source, masks, keep, iou, sums, best, want_winners: bool):
        self.source, self.masks, self.keep, self.iou, self.row_sums, self.best = source, masks, keep, iou, sums, best
        B, M = keep.shape
        self._host = torch.empty((B, M + 1), dtype=torch.int32, pin_memory=True)
        self._host.copy_(torch.cat([keep, best[:, None]], dim=1), non_blocking=True)
        self.winners = self._winners_host = None
        if want_winners:  # True: on the device and in page-locked host memory; "device": on the device only
            self.winners = masks[torch.arange(B, device=masks.device), best.long().clamp_(min=0)]
            if want_winners != "device":
                self._winners_host = torch.empty(self.winners.shape, dtype=torch.uint8, pin_memory=True)
                self._winners_host.copy_(self.winners, non_blocking=True)
        self._done = torch.cuda.Event()
        self._done.record(torch.cuda.current_stream(masks.device))

    def _rows(self):
        self._done.synchronize()
        rows = self._host.tolist()
        if any(r[-1] < 0 for r in rows):
            raise RuntimeError("sm_vote_masks_batch_u8 returned no winner")
        return rows

    def winners_host(self) -> torch.Tensor:
        assert self.winners is not None, "vote_mask_batch_async(..., winners=True)"
        self._rows()
        if self._winners_host is None:  # asked for on the device only: copy now
            self._winners_host = self.winners.cpu()
        return self._winners_host

    def result(self):
        out = []
        M = self.keep.shape[1]
        for b, row in enumerate(self._rows()):
            keep_h, best_h = row[:M], row[M]
            new_to_prev = {}
            for prev, k in enumerate(keep_h):
                if k:
                    new_to_prev[len(new_to_prev)] = prev
            out.append((self.source[b, best_h], {v: k for k, v in new_to_prev.items()}[best_h], new_to_prev))
        return out


class PendingRuns:
    """Run boundaries of a batch of masks (``rle_runs_async``), queued on a stream: ``result()`` waits for the batch's two small copies
    and returns one COCO uncompressed run-length dict per mask ({"size": [H, W], "counts": [...]}: column-major runs, zeros first)."""

    def __init__(self, masks: torch.Tensor, cap: int, sizes=None):
        B, H, W = masks.shape
        self.masks, self.cap = masks, cap
        self.sizes = [(int(H), int(W))] * B if sizes is None else [(int(h), int(w)) for h, w in sizes]
        dev = masks.device
        self._starts = torch.empty((B, cap), dtype=torch.int32, device=dev)
        self._info = torch.empty((B, 2), dtype=torch.int32, device=dev)
        self._sizes_dev = None if sizes is None else _sizes_tensor(self.sizes, (H, W), dev)
        N.check(N.load().sm_rle_runs_u8(masks.data_ptr(), B, H, W, None if sizes is None else self._sizes_dev.data_ptr(),
                                        self._starts.data_ptr(), cap, self._info.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                "sm_rle_runs_u8")
        self._info_h = torch.empty((B, 2), dtype=torch.int32, pin_memory=True)
        self._info_h.copy_(self._info, non_blocking=True)
        self._done = torch.cuda.Event()
        self._done.record(torch.cuda.current_stream(dev))

    def result(self):
        import numpy as np
        self._done.synchronize()
        info = self._info_h.numpy()
        longest = int(info[:, 0].max(initial=0))
        if longest > self.cap:  # more runs than the buffer holds (noise-like masks): once more on the device with room for the longest
            again = PendingRuns(self.masks, longest, None if self._sizes_dev is None else self.sizes)
            return again.result()
        starts = self._starts[:, :max(longest, 1)].cpu().numpy()  # only as many columns as the longest code needs
        out = []
        for b in range(info.shape[0]):
            n, first = int(info[b, 0]), int(info[b, 1])
            H, W = self.sizes[b]
            bounds = np.concatenate([[0], starts[b, :n], [H * W]])
            counts = np.diff(bounds).tolist()
            out.append({"size": [H, W], "counts": ([0] + counts) if first else counts})
        return out


def _sizes_tensor(sizes, plane, dev) -> torch.Tensor:
    """[(H_b, W_b)] -> (B, 2) int32 on the device, checked against the planes they are cut from"""
    t = torch.tensor(sizes, dtype=torch.int32).reshape(-1, 2)
    assert int(t[:, 0].min()) >= 1 and int(t[:, 1].min()) >= 1 and int(t[:, 0].max()) <= plane[0] and int(t[:, 1].max()) <= plane[1], \
        f"sizes must lie inside the {plane[0]}x{plane[1]} planes"
    return t.to(dev, non_blocking=True)


def rle_runs_async(masks: torch.Tensor, cap: int = 8192, sizes=None) -> PendingRuns:
    """masks (B, H, W) uint8 on a HIP device (non-zero = set) -> their run-length codes, without waiting (``.result()``).  What
    ``mask_generator.rle_encode`` computes on the host, from the run boundaries the device finds (``sm_rle_runs_u8``).  ``sizes``
    [(H_b, W_b)]: image b is the top-left H_b x W_b of its plane."""
    if not masks.is_cuda:
        raise RuntimeError("rle_runs_async (MI355X) needs its masks on a HIP device; mask_generator.rle_encode is the host form")
    m = masks.to(torch.uint8).contiguous()
    assert m.dim() == 3
    return PendingRuns(m, int(min(cap, max(1, m.shape[1] * m.shape[2]))), sizes)


def vote_mask_batch_async(batch_pred_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False,
                          winners: bool = False, sizes=None) -> PendingVotes:
    """``vote_mask`` for B images in one launch sequence on the current stream, without waiting for it.  ``sizes`` [(H_b, W_b)]: the
    images have different sizes inside planes of one size (padded to one token grid): image b's candidates are the top-left
    H_b x W_b of its planes, and its result is that of ``vote_mask`` on those crops."""
    if not batch_pred_masks.is_cuda:
        raise RuntimeError("vote_mask_batch (MI355X) needs its candidates on a HIP device; there is no CPU fallback")
    m = batch_pred_masks.to(torch.uint8).contiguous()
    B, M, H, W = m.shape
    lib = N.load()
    dev = m.device
    nbytes = lib.sm_vote_workspace_bytes(M, H, W)
    if nbytes == 0:
        raise ValueError(f"vote_mask_batch: {M} candidates of {H}x{W} (1..64 candidates)")
    ws = _arena("vote", nbytes * B, dev)
    keep = torch.empty((B, M), dtype=torch.int32, device=dev)
    iou = torch.empty((B, M, M), dtype=torch.float32, device=dev)
    sums = torch.empty((B, M), dtype=torch.float32, device=dev)
    best = torch.empty(B, dtype=torch.int32, device=dev)
    sz = None if sizes is None else _sizes_tensor([(int(h), int(w)) for h, w in sizes], (H, W), dev)
    assert sz is None or sz.shape[0] == B
    N.check(lib.sm_vote_masks_sized_u8(m.data_ptr(), B, M, H, W, None if sz is None else sz.data_ptr(), int(remove_long_masks),
                                       int(remove_small_large_masks), keep.data_ptr(), iou.data_ptr(), sums.data_ptr(), best.data_ptr(),
                                       ws.data_ptr(), nbytes * B, torch.cuda.current_stream(dev).cuda_stream), "sm_vote_masks_sized_u8")
    pend = PendingVotes(batch_pred_masks, m, keep, iou, sums, best, winners)
    pend.sizes_dev = sz  # (kept alive until the launches have run)
    return pend


def vote_mask_batch(batch_pred_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False):
    """``vote_mask`` for B images of one size in one launch sequence and ONE device-to-host copy: (B, M, H, W) 0/1 -> list of B
    tuples (best mask (H, W), best index among the survivors, new_index_to_prev_index), each what ``vote_mask`` returns for that image."""
    pend = vote_mask_batch_async(batch_pred_masks, remove_long_masks, remove_small_large_masks)
    vote_mask_batch.last = {"keep": pend.keep, "iou": pend.iou, "row_sums": pend.row_sums, "best": pend.best}
    return pend.result()


def kmeans(features: torch.Tensor, k: int, iters: int = 20):
    """features (B, n, 384) fp32 on a HIP device -> labels (B, n) int32: Lloyd's k-means with farthest-point initial centres,
    deterministic (csrc/cluster.hip).  Also returns the centres (B, k, 384)."""
    if not features.is_cuda:
        raise RuntimeError("kmeans (MI355X) needs its features on a HIP device; there is no CPU fallback")
    f = features.contiguous().float()
    B, n, d = f.shape
    assert d == N.EMBED
    labels = torch.empty((B, n), dtype=torch.int32, device=f.device)
    centers = torch.empty((B, k, d), dtype=torch.float32, device=f.device)
    ws = torch.empty((B, n), dtype=torch.float32, device=f.device)
    N.check(N.load().sm_kmeans_f32(f.data_ptr(), B, n, k, iters, labels.data_ptr(), centers.data_ptr(), ws.data_ptr(),
                                   torch.cuda.current_stream(f.device).cuda_stream), "sm_kmeans_f32")
    return labels, centers


def spectral_cluster(features: torch.Tensor, cluster_sizes: Sequence[int] = (2, 3, 4), n_neighbors: int = 10, tol: float = 1e-9,
                     degree: int = 24, max_outer: int = 60, return_details: bool = False):
    """features (B, n, 384) fp32 on a HIP device -> labels (B, len(cluster_sizes), n) int32: normalised spectral clustering
    (csrc/spectral.hip), one eigen-solve per image shared by every cluster size.  ``return_details`` adds a dict with the
    neighbour lists ``knn`` (B, n, n_neighbors - 1), the ``eigenvalues`` (B, kw) of the normalised Laplacian, the ``embedding``
    (B, n, kw), the eigen-``residuals`` (B, kw) and ``info`` (B, 4: outer iterations, block mat-vecs, converged, guard)."""
    if not features.is_cuda:
        raise RuntimeError("spectral_cluster (MI355X) needs its features on a HIP device; there is no CPU fallback")
    f = features.contiguous().float()
    B, n, d = f.shape
    assert d == N.EMBED
    sizes = [int(k) for k in cluster_sizes]
    kw = max(sizes)
    lib = N.load()
    nbytes = lib.sm_spectral_workspace_bytes(B, n, n_neighbors, kw)
    if nbytes == 0:
        raise ValueError(f"spectral_cluster: {n} points, n_neighbors {n_neighbors}, {kw} vectors (16 <= n <= 8192, n_neighbors 2..33, k <= 6)")
    dev = f.device
    ws = _arena("spectral", nbytes, dev)
    labels = torch.empty((B, len(sizes), n), dtype=torch.int32, device=dev)
    a = N.SpectralArgs()
    a.features, a.labels = f.data_ptr(), labels.data_ptr()
    host_sizes = (C.c_int32 * len(sizes))(*sizes)
    a.cluster_sizes = C.cast(host_sizes, C.POINTER(C.c_int32))
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    a.tol, a.B, a.n, a.n_sizes, a.n_neighbors, a.degree, a.max_outer = tol, B, n, len(sizes), n_neighbors, degree, max_outer
    det = None
    if return_details:
        m = min(n_neighbors - 1, n - 1)
        det = {"knn": torch.empty((B, n, m), dtype=torch.int32, device=dev),
               "eigenvalues": torch.empty((B, kw), dtype=torch.float64, device=dev),
               "embedding": torch.empty((B, n, kw), dtype=torch.float64, device=dev),
               "residuals": torch.empty((B, kw), dtype=torch.float64, device=dev),
               "info": torch.empty((B, 4), dtype=torch.int32, device=dev)}
        a.knn, a.eigenvalues, a.embedding = det["knn"].data_ptr(), det["eigenvalues"].data_ptr(), det["embedding"].data_ptr()
        a.residuals, a.info = det["residuals"].data_ptr(), det["info"].data_ptr()
    N.check(lib.sm_spectral_cluster_f32(C.byref(a), torch.cuda.current_stream(dev).cuda_stream), "sm_spectral_cluster_f32")
    labels.record_stream(torch.cuda.current_stream(dev))
    return (labels, det) if return_details else labels


def upsample_tokens_aligned(tokens: torch.Tensor, gh: int, gw: int, scale: int = 2, scratch: bool = False) -> torch.Tensor:
    """tokens (B, gh*gw, 384) -> (B, scale*gh, scale*gw, 384): F.interpolate(scale_factor=scale, mode="bilinear", align_corners=True)."""
    t = tokens.contiguous().float()
    B = t.shape[0]
    shape = (B, scale * gh, scale * gw, N.EMBED)
    # ``scratch``: the result lives in the stream's arena (valid until the next such call on this stream) - extract_candidate_masks
    up = (_arena("features", 4 * B * scale * gh * scale * gw * N.EMBED, t.device).view(torch.float32).view(shape) if scratch
          else torch.empty(shape, dtype=torch.float32, device=t.device))
    N.check(N.load().sm_upsample_tokens_aligned_f32(t.data_ptr(), t.stride(0), up.data_ptr(), B, gh, gw, scale,
                                                    torch.cuda.current_stream(t.device).cuda_stream), "sm_upsample_tokens_aligned_f32")
    return up


def labels_to_masks(labels: torch.Tensor, k: int, scale: int, H: int, W: int) -> torch.Tensor:
    """labels (lh, lw) int32 -> (k, H, W) uint8: one-hot (utils/misc.py:10-35) + nearest up-sample by ``scale`` + crop."""
    lab = labels.contiguous().to(torch.int32)
    lh, lw = lab.shape
    masks = torch.empty((k, H, W), dtype=torch.uint8, device=lab.device)
    N.check(N.load().sm_labels_to_masks_u8(lab.data_ptr(), lh, lw, scale, k, H, W, masks.data_ptr(),
                                           torch.cuda.current_stream(lab.device).cuda_stream), "sm_labels_to_masks_u8")
    return masks


def labels_to_masks_batch(labels: torch.Tensor, cluster_sizes: Sequence[int], lh: int, lw: int, scale: int, H: int, W: int) -> torch.Tensor:
    """labels (B, len(cluster_sizes), lh * lw) int32 -> (B, sum(cluster_sizes), H, W) uint8: ``labels_to_masks`` for every (image,
    cluster size) in one launch."""
    lab = labels.contiguous().to(torch.int32)
    B, ns = lab.shape[:2]
    sizes = [int(k) for k in cluster_sizes]
    assert ns == len(sizes) and lab.shape[2] == lh * lw
    masks = torch.empty((B, sum(sizes), H, W), dtype=torch.uint8, device=lab.device)
    host_sizes = (C.c_int32 * ns)(*sizes)
    N.check(N.load().sm_labels_to_masks_batch_u8(lab.data_ptr(), B, ns, C.cast(host_sizes, C.POINTER(C.c_int32)), lh, lw, scale, H, W,
                                                 masks.data_ptr(), torch.cuda.current_stream(lab.device).cuda_stream),
            "sm_labels_to_masks_batch_u8")
    return masks


@torch.no_grad()
def extract_candidate_masks(model, x: torch.Tensor, cluster_sizes=(2, 3, 4), clusterer=None, iters: int = 20,
                            cluster_type: str = "spectral", n_neighbors: int = 10) -> torch.Tensor:
    """mask_generator.pyc@L136-200, DINO branch, for normalised images x (B, 3, H, W) of one size on a HIP device (the reference's
    DataLoader runs batch 1; images are independent, so a batch is the same thing B times): layer-12 patch tokens (the image
    zero-padded to a patch multiple) -> bilinear x2 (align_corners=True) -> ``clusterer(features (B, n, 384), k)`` -> labels ->
    one-hot -> nearest up-sample by patch // 2 -> crop to (H, W).  Returns (sum(cluster_sizes), H, W) uint8 for B = 1 - the
    candidates ``vote_mask`` takes - and (B, sum(cluster_sizes), H, W) otherwise.  ``cluster_type``: "spectral" (default, as the
    shipped YAML and ``MaskGenerator.__init__``: ``spectral_cluster``) or "k-means" (the reference's spelling, ``mask_generator.pyc@L30-38``;
    "kmeans" is accepted too); ``clusterer`` overrides both with a
    callable ``(features, k) -> labels (B, n)``.  Both built-in clusterers are parity UNPINNED (module header)."""
    assert x.dim() == 4
    assert cluster_type in CLUSTER_TYPES + ("kmeans",), cluster_type  # "kmeans": alias of the reference's "k-means"
    B, _, H, W = x.shape
    p = model.encoder.patch_size
    tok = model(x, encoder_only=True)["patch_tokens"]  # (B, gh, gw, 384), final-normed, cls dropped
    gh, gw = tok.shape[1:3]
    feats = upsample_tokens_aligned(tok.reshape(B, gh * gw, N.EMBED), gh, gw, 2, scratch=True)  # (B, 2gh, 2gw, 384)
    flat = feats.reshape(B, 4 * gh * gw, N.EMBED)
    if clusterer is not None:
        lab = torch.stack([clusterer(flat, k).reshape(B, 4 * gh * gw).to(torch.int32) for k in cluster_sizes], dim=1)
    elif cluster_type == "spectral":  # mask_generator.pyc@L30-38: "k-means" -> KMeansClustering, anything else -> SpectralClustering
        lab = spectral_cluster(flat, cluster_sizes, n_neighbors)
    else:
        lab = torch.stack([kmeans(flat, k, iters)[0] for k in cluster_sizes], dim=1)
    out = labels_to_masks_batch(lab, cluster_sizes, 2 * gh, 2 * gw, p // 2, H, W)  # (B, sum k, H, W): one launch
    return out[0] if B == 1 else out
