"""Mirror of the reference's pseudo-mask generator, DINO branch (datasets/mask_generator, bytecode only: SURVEY.md Appendix B):
``extract_candidate_masks`` (@L136-200: encoder tokens -> bilinear x2 ``align_corners=True`` -> ``clusterer(features, k)`` for
k in {2, 3, 4} -> one-hot -> nearest up-sample to the image) and ``vote_mask`` (@L202-230) with ``utils.misc.filter_masks``
(utils/misc.py:285-314), all on the MI355X.

The reference's ``clusterings`` module (``KMeansClustering`` / ``SpectralClustering``: faiss k-NN affinity + eigen-decomposition)
does not exist in its repository in any form.  ``kmeans`` below is a stated stand-in for its ``cluster_type="kmeans"`` option
(parity UNPINNED, csrc/cluster.hip); any other clusterer can be passed as a callable, as the reference's class takes one.  The
ResNet-50 MoCo-v2 / SwAV feature branches are out of scope."""
from typing import Dict, Tuple

import torch

from . import _native as N


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


def upsample_tokens_aligned(tokens: torch.Tensor, gh: int, gw: int, scale: int = 2) -> torch.Tensor:
    """tokens (B, gh*gw, 384) -> (B, scale*gh, scale*gw, 384): F.interpolate(scale_factor=scale, mode="bilinear", align_corners=True)."""
    t = tokens.contiguous().float()
    B = t.shape[0]
    up = torch.empty((B, scale * gh, scale * gw, N.EMBED), dtype=torch.float32, device=t.device)
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


@torch.no_grad()
def extract_candidate_masks(model, x: torch.Tensor, cluster_sizes=(2, 3, 4), clusterer=None, iters: int = 20) -> torch.Tensor:
    """mask_generator.pyc@L136-200, DINO branch, for ONE normalised image x (1, 3, H, W) on a HIP device: layer-12 patch tokens
    (the image zero-padded to a patch multiple) -> bilinear x2 (align_corners=True) -> ``clusterer(features (1, n, 384), k)`` ->
    labels -> one-hot -> nearest up-sample by patch // 2 -> crop to (H, W).  Returns (sum(cluster_sizes), H, W) uint8, the
    candidates ``vote_mask`` takes.  ``clusterer`` defaults to the device k-means above (stand-in, parity unpinned)."""
    assert x.dim() == 4 and x.shape[0] == 1, "one image at a time, as the reference's DataLoader(batch_size=1)"
    H, W = x.shape[-2:]
    p = model.encoder.patch_size
    tok = model(x, encoder_only=True)["patch_tokens"]  # (1, gh, gw, 384), final-normed, cls dropped
    gh, gw = tok.shape[1:3]
    feats = upsample_tokens_aligned(tok.reshape(1, gh * gw, N.EMBED), gh, gw, 2)  # (1, 2gh, 2gw, 384)
    flat = feats.reshape(1, 4 * gh * gw, N.EMBED)
    out = []
    for k in cluster_sizes:
        labels = (clusterer(flat, k) if clusterer is not None else kmeans(flat, k, iters)[0]).reshape(2 * gh, 2 * gw)
        out.append(labels_to_masks(labels, k, p // 2, H, W))
    return torch.cat(out, dim=0)
