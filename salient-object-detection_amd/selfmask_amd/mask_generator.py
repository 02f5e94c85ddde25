"""Mirror of the reference's ``datasets.mask_generator.MaskGenerator`` (bytecode only: SURVEY.md Appendix B, mask_generator.pyc@L21-252;
its constants are pinned by tests/golden/evaluator_constants.json), DINO branch, on the MI355X:

    MaskGenerator(cluster_sizes=(2, 3, 4), cluster_type="spectral", feature_types=["dino"], network=model)(p_images)
        -> {filename: run-length-encoded salient mask}

``extract_candidate_masks`` (@L136-200): every image at its native resolution, normalised as ``CustomDataset`` does
(/root/reference/datasets/custom_dataset.py:26-32: RGB, ``to_tensor``, ImageNet mean / std), zero-padded to a multiple of the stride
(``pad_input_image`` @L124-134 = what ``make_input_divisible`` does inside the encoder), layer-12 tokens -> bilinear x2
(``align_corners=True``) -> ``clusterer(features, k)`` for k in cluster_sizes -> one-hot -> nearest up-sample -> crop.  Images of one size
share a batch (the reference's DataLoader runs batch 1; images are independent).  ``vote_mask`` (@L202-230) picks the candidate that
agrees most with the others.  ``__call__`` (@L232-252) returns the winner per file name, run-length encoded.

Differences kept on purpose: (1) only the ``"dino"`` feature type - the MoCo-v2 / SwAV ResNet-50 branches need backbones and weights that
are absent from the reference repository, asking for them raises; (2) the encoder is the ``network`` handed in
(a ``selfmask_amd.MaskFormer`` whose encoder holds the DINO weights): the reference fetches them from the network at construction time
(utils/misc.py:196,243), which must never happen here; (3) the reference encodes with ``pycocotools.mask.encode`` (absent from this
image): ``rle_encode`` writes COCO's UNCOMPRESSED run-length form ({"size": [h, w], "counts": [n0, n1, ...]}, column-major, first run =
zeros), which ``pycocotools.mask.frPyObjects`` turns into the compressed string; (4) both clusterers are parity UNPINNED - the
reference's ``clusterings`` module is absent in every form (voting.py)."""
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import voting as VT
from .datasets import MEAN, STD


def rle_encode(mask: np.ndarray) -> dict:
    """(H, W) 0/1 -> COCO uncompressed RLE: runs over the column-major (Fortran) order, starting with the zeros."""
    m = np.asarray(mask).astype(bool)
    flat = m.flatten(order="F")
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(bounds).tolist()
    if flat.size and flat[0]:
        counts = [0] + counts
    return {"size": [int(m.shape[0]), int(m.shape[1])], "counts": [int(c) for c in counts]}


def rle_decode(rle: dict) -> np.ndarray:
    h, w = rle["size"]
    vals = np.zeros(len(rle["counts"]), bool)
    vals[1::2] = True
    return np.repeat(vals, rle["counts"]).reshape((h, w), order="F").astype(np.uint8)


class MaskGenerator:
    def __init__(self, cluster_sizes: Sequence[int] = VT.DEFAULT_CLUSTER_SIZES, cluster_type: str = VT.DEFAULT_CLUSTER_TYPE,
                 feature_types: Sequence[str] = ("dino",), use_gpu: bool = True, device: torch.device = torch.device("cuda:0"),
                 network=None, batch_size: int = 16, n_neighbors: int = 10):
        assert cluster_type in VT.CLUSTER_TYPES + ("kmeans",), cluster_type  # mask_generator.pyc@L30: ('k-means', 'spectral')
        unsupported = [f for f in feature_types if f != "dino"]
        if unsupported:
            raise NotImplementedError(f"feature types {unsupported}: the MoCo-v2 / SwAV ResNet-50 backbones and their weights are absent "
                                      f"from the reference repository; only 'dino' is built")
        if network is None:
            raise ValueError("MaskGenerator needs `network` (a selfmask_amd.MaskFormer whose encoder holds the DINO ViT-S weights): the "
                             "reference downloads them at construction time, which this build never does")
        if not use_gpu:
            raise RuntimeError("the MI355X build has no CPU path")
        self.cluster_sizes, self.cluster_type, self.feature_types = tuple(int(k) for k in cluster_sizes), cluster_type, list(feature_types)
        self.device, self.network, self.batch_size, self.n_neighbors = torch.device(device), network, int(batch_size), int(n_neighbors)

    # ---- mask_generator.pyc@L136-200 --------------------------------------------------------------------------------------------
    def _load(self, p_image: str) -> torch.Tensor:
        from PIL import Image
        rgb = np.asarray(Image.open(p_image).convert("RGB"), np.float32) / np.float32(255.0)  # to_tensor
        x = (rgb - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)               # normalize
        return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))

    @torch.no_grad()
    def extract_candidate_masks(self, p_images: Sequence[str]) -> Dict[str, torch.Tensor]:
        """file name -> (sum(cluster_sizes), H, W) uint8 on the device (the reference concatenates the candidates of its three
        feature types per file name; here there is one)."""
        by_size = defaultdict(list)
        tensors = {}
        for p in p_images:
            x = self._load(p)
            tensors[p] = x
            by_size[tuple(x.shape[-2:])].append(p)
        out: Dict[str, torch.Tensor] = {}
        for (_h, _w), paths in by_size.items():
            for s in range(0, len(paths), self.batch_size):
                chunk = paths[s:s + self.batch_size]
                x = torch.stack([tensors[p] for p in chunk]).to(self.device)
                cands = VT.extract_candidate_masks(self.network, x, self.cluster_sizes, cluster_type=self.cluster_type,
                                                   n_neighbors=self.n_neighbors)
                cands = cands[None] if cands.dim() == 3 else cands
                for p, c in zip(chunk, cands):
                    out[p.split("/")[-1]] = c
        return out

    # ---- mask_generator.pyc@L202-230 --------------------------------------------------------------------------------------------
    def vote_mask(self, batch_pred_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False):
        return VT.vote_mask(batch_pred_masks, remove_long_masks, remove_small_large_masks)

    # ---- mask_generator.pyc@L232-252 --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, p_images: Sequence[str], remove_long_masks: bool = True, remove_small_large_masks: bool = False,
                 encode: Optional[bool] = True) -> Dict[str, object]:
        cands = self.extract_candidate_masks(p_images)
        by_shape = defaultdict(list)
        for name, c in cands.items():
            by_shape[tuple(c.shape)].append(name)
        result: Dict[str, object] = {}
        for _shape, names in by_shape.items():  # one vote launch sequence and one device-to-host copy per group of one size
            votes = VT.vote_mask_batch(torch.stack([cands[n] for n in names]), remove_long_masks, remove_small_large_masks)
            for n, (best_mask, _best, _map) in zip(names, votes):
                m = best_mask.cpu().numpy()
                result[n] = rle_encode(m) if encode else m
        return result
