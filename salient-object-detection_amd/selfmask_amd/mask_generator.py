"""Mirror of the reference's ``datasets.mask_generator.MaskGenerator`` (bytecode only: SURVEY.md Appendix B, mask_generator.pyc@L21-252;
its constants are pinned by tests/golden/evaluator_constants.json), DINO branch, on the MI355X:

    MaskGenerator(cluster_sizes=(2, 3, 4), cluster_type="spectral", feature_types=["dino"], network=model)(p_images)
        -> {filename: run-length-encoded salient mask}

``extract_candidate_masks`` (@L136-200): every image at its native resolution, normalised as ``CustomDataset`` does
(/root/reference/datasets/custom_dataset.py:26-32: RGB, ``to_tensor``, ImageNet mean / std), zero-padded to a multiple of the stride
(``pad_input_image`` @L124-134 = what ``make_input_divisible`` does inside the encoder), layer-12 tokens -> bilinear x2
(``align_corners=True``) -> ``clusterer(features, k)`` for k in cluster_sizes -> one-hot -> nearest up-sample -> crop.  Images that pad to
one patch grid share a batch (the reference's DataLoader runs batch 1; images are independent, and an image's padded input is the same
in a batch as alone): the crop to its own H x W happens inside the vote and the run-length kernels.  ``vote_mask`` (@L202-230) picks the candidate that
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


class _Files:
    """what PrefetchingLoader needs of a dataset: image paths, no ground truth"""

    def __init__(self, p_imgs):
        self.p_imgs, self.p_gts = list(p_imgs), [None] * len(p_imgs)


class MaskGenerator:
    def __init__(self, cluster_sizes: Sequence[int] = VT.DEFAULT_CLUSTER_SIZES, cluster_type: str = VT.DEFAULT_CLUSTER_TYPE,
                 feature_types: Sequence[str] = ("dino",), use_gpu: bool = True, device: torch.device = torch.device("cuda:0"),
                 network=None, batch_size: int = 128, n_neighbors: int = 10, streams: int = 3, workers: Optional[int] = None):
        """``batch_size``: the most images of ONE size clustered together - the eigen-solver runs one workgroup per image, so a launch
        takes about as long for 128 images as for 8 (profiles/r04_pseudo_masks_by_batch.log); ``streams``: batches in flight (one's
        eigen-solve, half of the CUs at 128 images, runs beside the next one's encoder); ``workers``: decode processes
        (default: this rank's share of the host cores)."""
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
        self.streams, self.workers = max(1, int(streams)), workers
        self._ring = None  # the streams of __call__, made once: the per-stream scratch of the clusterer (voting._arena) stays bounded

    # ---- mask_generator.pyc@L136-200 --------------------------------------------------------------------------------------------
    def _load(self, p_image: str) -> torch.Tensor:
        """one file as ``CustomDataset`` prepares it on the host (custom_dataset.py:26-32) - what the batches below hold, per image
        (tests compare the two)"""
        from PIL import Image
        rgb = np.asarray(Image.open(p_image).convert("RGB"), np.float32) / np.float32(255.0)  # to_tensor
        x = (rgb - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)               # normalize
        return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))

    def _batches(self, p_images: Sequence[str], pack: bool = False):
        """-> (file names, decoded uint8 RGB arrays) per batch of at most ``batch_size`` images that pad to ONE patch grid
        (``pipeline.native_buckets``: the evaluator's native-resolution buckets): the headers give the sizes, the decode processes of the
        input pipeline (decode_pool.py) the pixels, a few batches ahead of the device.  Largest buckets first.  ``pack``: the second item
        is (page-locked staging buffers of the batch, [(H, W)]) assembled on the loader's packing thread - what ``__call__`` consumes."""
        from .datasets import probe_size
        from .pipeline import PrefetchingLoader, native_buckets
        p_images = list(p_images)
        sizes = [probe_size(p) for p in p_images]  # headers only
        batches = sorted(native_buckets(sizes, self.network.encoder.patch_size, self.batch_size), key=len, reverse=True)
        loader = PrefetchingLoader(_Files(p_images), range(len(p_images)), self.batch_size, workers=self.workers, batches=batches, pack=pack)
        for rgbs, _gts, idx in loader:
            yield [p_images[i].split("/")[-1] for i in idx], rgbs

    def _candidates(self, rgbs):
        """decoded images of one patch grid (arrays, or the loader's (staging buffers, shapes)) -> ((B, sum(cluster_sizes), Hp, Wp) uint8
        candidates, [(H_b, W_b)]), queued on the current stream: every image zero-padded to the grid's Hp x Wp after normalisation - what
        ``pad_input_image`` (@L124-134) builds for it alone - and its candidates are the top-left H_b x W_b of its planes"""
        from .pipeline import preprocess_on_device
        P = self.network.encoder.patch_size
        packed = None
        if isinstance(rgbs, tuple):
            packed, rgbs = rgbs
        sizes = [(int(r[0]), int(r[1])) if packed is not None else (int(r.shape[0]), int(r.shape[1])) for r in rgbs]
        Hp, Wp = -(-max(h for h, _ in sizes) // P) * P, -(-max(w for _, w in sizes) // P) * P
        x = preprocess_on_device(rgbs, None, self.device, pinned=True, packed=packed, pad_to=(Hp, Wp))  # to_tensor + normalize + pad (device)
        cands = VT.extract_candidate_masks(self.network, x, self.cluster_sizes, cluster_type=self.cluster_type, n_neighbors=self.n_neighbors)
        return (cands[None] if cands.dim() == 3 else cands), sizes

    @torch.no_grad()
    def extract_candidate_masks(self, p_images: Sequence[str]) -> Dict[str, torch.Tensor]:
        """file name -> (sum(cluster_sizes), H, W) uint8 on the device (the reference concatenates the candidates of its three
        feature types per file name; here there is one)."""
        out: Dict[str, torch.Tensor] = {}
        for names, rgbs in self._batches(p_images):
            cands, sizes = self._candidates(rgbs)
            for n, c, (h, w) in zip(names, cands, sizes):
                out[n] = c[:, :h, :w]
        return out

    # ---- mask_generator.pyc@L202-230 --------------------------------------------------------------------------------------------
    def vote_mask(self, batch_pred_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False):
        return VT.vote_mask(batch_pred_masks, remove_long_masks, remove_small_large_masks)

    # ---- mask_generator.pyc@L232-252 --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, p_images: Sequence[str], remove_long_masks: bool = True, remove_small_large_masks: bool = False,
                 encode: Optional[bool] = True, comm=None) -> Dict[str, object]:
        """Candidates and vote of one batch are queued on one stream of a ring and read back ``streams`` batches later: images are
        independent, so the result per file is what the reference's extract-everything-then-vote order gives.  ``comm`` (a
        ``distributed.TorchDistComm``): this rank takes files rank, rank + W, ... of the list (``shard_indices``, as the evaluator
        shards its images), and every rank returns the codes of ALL files (one all-gather of the encoded shards at the end)."""
        if comm is not None and comm.world_size > 1:
            from .distributed import gather_dicts, shard_indices
            assert encode, "the gather exchanges run-length codes"
            p_images = list(p_images)
            names = [p.split("/")[-1] for p in p_images]
            assert len(set(names)) == len(names), "file names must be unique (they key the result)"
            mine = self([p_images[i] for i in shard_indices(len(p_images), comm.rank, comm.world_size)], remove_long_masks,
                        remove_small_large_masks, True)
            merged = gather_dicts(mine, comm, self.device)
            return {n: merged[n] for n in names}  # the list's order, whatever the sharding
        from collections import deque
        from .streams import StreamRing
        if self._ring is None:
            self._ring = StreamRing(self.device, self.streams)
        ring = self._ring
        ring.home = torch.cuda.current_stream(self.device)  # whatever the caller queued so far is visible to the ring's streams
        ring.fork()
        pending = deque()
        result: Dict[str, object] = {}

        def settle():
            names, votes, runs = pending.popleft()
            names, sizes = names
            if runs is not None:  # the run boundaries were found on the device: two small copies per batch instead of the masks
                votes.result()    # (raises if a batch came back without a winner)
                result.update(zip(names, runs.result()))
            else:
                for n, m, (h, w) in zip(names, votes.winners_host().numpy(), sizes):
                    result[n] = m[:h, :w].copy()

        for names, rgbs in self._batches(p_images, pack=True):
            with ring.next():
                cands, sizes = self._candidates(rgbs)
                votes = VT.vote_mask_batch_async(cands, remove_long_masks, remove_small_large_masks, winners="device" if encode else True,
                                                 sizes=sizes)
                pending.append(((names, sizes), votes, VT.rle_runs_async(votes.winners, sizes=sizes) if encode else None))
            if len(pending) >= len(ring.streams):
                settle()
        while pending:
            settle()
        ring.join()
        return result
