"""Mirror of the reference's ``evaluator.Evaluator`` (bytecode-only in the reference; behaviour per SURVEY.md
Appendix A, evaluator.pyc@L17-373) with the per-image work on the MI355X:

  forward (HIP) -> last decoder layer -> bilinear up-sample + crop -> upper-bound query by IoU, arg-max-objectness
  query -> 7 metrics x {pick, upper bound} (sm_evaluate_masks_f32) -> running means -> ``metrics_<dataset>.txt``.

Two operating points: the reference's own (batch 1, native resolution, up-sample factor patch_size // scale_factor,
i.e. the hard-coded 4 of evaluator.pyc@L209-211 for ViT-S/8) and a batched mode (``img_size`` given: inputs resized
to S x S, masks resized to each GT's native size).  Images shard across ranks (distributed.py).
"""
import argparse
import os
from typing import Optional

import numpy as np
import torch

from . import ops
from .base_structure import BaseStructure
from .datasets import get_dataset
from .graphs import GraphedForward
from .streams import DEFAULT_STREAMS, StreamRing
from .distributed import HEADER, KEYS, SingleComm, TorchDistComm, average_rows, gather_rows, shard_indices


# constants of evaluator.pyc@L164-309 (read from the bytecode as data: tests/golden/evaluator_constants.json pins them)
REFERENCE_UPSAMPLE = {"scale_factor": 4, "mode": "bilinear", "align_corners": False}  # @L209-211: patch 8 // scale_factor 2
MASK_THRESHOLD = 0.5          # @L216: pred_masks > 0.5 before the upper-bound search
DATALOADER_WORKERS = 4        # @L174 (the product decodes on its own worker pool)
VISUALISE_EVERY = 250         # @L230: batches between visualiser dumps (the visualiser is out of scope)


class Evaluator(BaseStructure):
    def __init__(self, network: callable, arch: str = "vit_small", dir_dataset: str = "datasets",
                 visualizer: Optional[callable] = None, debug: bool = False):
        super().__init__(model=network, visualizer=visualizer)
        assert os.path.exists(dir_dataset), dir_dataset  # evaluator.pyc@L27
        self.arch, self.debug, self.dir_dataset, self.network = arch, debug, dir_dataset, network

    @torch.no_grad()
    def __call__(self, dataset_name: str, dir_ckpt: str, img_size: Optional[int] = None, scale_factor: int = 2,
                 batch_size: int = 1, device: torch.device = torch.device("cuda:0"), cost_type: str = "iou",
                 comm=None, streams: int = DEFAULT_STREAMS, hip_graph: bool = True, input_pipeline: str = "device",
                 workers: Optional[int] = None, refine: Optional[str] = None) -> dict:
        """evaluator.pyc@L164-309.  Extra keyword arguments (not in the reference): ``comm`` / ``streams`` / ``hip_graph``
        (image sharding, batches in flight, graph replay) and ``input_pipeline``: "device" (default) decodes on a pool of
        ``workers`` host threads running ahead of the GPU and does resize + ToTensor + Normalize in HIP kernels
        (pipeline.py, bit-identical inputs), "host" is the reference's order of work: one thread, PIL + numpy per image.
        ``refine="bilateral"`` (batched mode only; BASELINE.json configs[2]): the picked query's mask is up-sampled to the
        S x S input, refined by ``bilateral_solver_output`` (bilateral_solver.py:152-193) against the resized RGB image -
        the whole batch in one launch sequence - and the solver's binary mask is scored by the same metric kernels; the
        returned dict gains the seven ``*_refined`` values and ``metrics_<dataset>_refined.txt`` is written."""
        assert cost_type == "iou", "the upper bound is chosen by IoU (evaluator.pyc@L216); other costs are unused"
        if not getattr(self.model, "use_binary_classifier", True):
            raise RuntimeError("the evaluator dereferences objectness unconditionally (evaluator.pyc@L219): "
                               "use_binary_classifier=True is required")
        if comm is None:
            import torch.distributed as dist
            comm = TorchDistComm() if dist.is_available() and dist.is_initialized() else SingleComm()
        dataset = get_dataset(self.dir_dataset, dataset_name, mode="test", eval_img_size=img_size)
        n_total = len(dataset)
        mine = shard_indices(n_total, comm.rank, comm.world_size)
        if self.debug:
            mine = mine[:batch_size]
        patch = self.model.encoder.patch_size
        scale = 0.0 if img_size is not None else float(patch // scale_factor)
        # Native resolution (the reference's own operating point, evaluator.pyc@L373: batch 1).  batch_size > 1 here batches the
        # images whose sizes pad to the SAME patch grid (pipeline.native_buckets): each image sits zero-padded in the top-left
        # corner of a (B, 3, gh P, gw P) batch - exactly the tensor make_input_divisible builds for it alone - and the forward
        # is batch-invariant bit for bit, so every result row equals the batch-1 row.
        bucketed = img_size is None and batch_size > 1
        plan = None
        if bucketed:
            from .pipeline import native_buckets
            plan = native_buckets([dataset.image_size(i) for i in mine], patch, batch_size)  # positions into `mine`
        if refine not in (None, "bilateral"):
            raise ValueError(f"refine={refine!r}: None or 'bilateral'")
        if refine and (img_size is None or input_pipeline != "device"):
            raise ValueError("refine='bilateral' runs in the batched mode (img_size given) on the device input pipeline")
        # One encoder attention path for the whole run (maskformer.attention_path): ragged last batches, token-grid buckets of
        # any size and shards of any world size then give the same bits per image as every other way of batching them
        prev_path = getattr(self.model, "attention_path", None)
        pin_path = ("fused" if (img_size is not None and batch_size >= 16) else "unfused") if prev_path == "auto" else None
        rows_local = torch.empty((len(mine), 16), dtype=torch.float32, device=device)
        rows_refined = torch.empty((len(mine), 16), dtype=torch.float32, device=device) if refine else None
        ring = StreamRing(device, streams)  # consecutive batches in flight on different HIP streams (streams.py)
        # recurring batch shapes replay one captured hipGraph per stream instead of ~150 launches (graphs.py)
        # native-resolution mode meets a new shape with almost every image: graphs would only thrash there.  In token-grid
        # buckets a shape is captured from its 32nd sighting per stream on: a capture costs ~10 ms, the bucketed run is
        # GPU-bound (its 0.6 ms of launches per forward are hidden), and ~50 grids on three streams rarely come back that often -
        # at batch 8 the admit-after-three policy spent 0.86 of 0.95 s capturing 86 graphs it replayed once or twice
        # (scripts/native_batch_sizes.py)
        self._graphed = GraphedForward(self.model, enabled=hip_graph and (img_size is not None or bucketed) and
                                       isinstance(self.model, torch.nn.Module), max_graphs=24 if bucketed else 8,
                                       admit_after=getattr(self, "bucket_graph_admit_after", 31) if bucketed else 2)
        assert input_pipeline in ("device", "host"), input_pipeline

        def padded(shapes):
            return (-(-max(h for h, _ in shapes) // patch) * patch, -(-max(w for _, w in shapes) // patch) * patch)

        def batches():
            if bucketed and input_pipeline == "host":
                for pos in plan:
                    items = [dataset[mine[p]] for p in pos]
                    Hp, Wp = padded([tuple(it["x"].shape[-2:]) for it in items])
                    xs = [torch.nn.functional.pad(it["x"], (0, Wp - it["x"].shape[-1], 0, Hp - it["x"].shape[-2])) for it in items]
                    yield pos, torch.stack(xs), [it["m"].squeeze() for it in items]
            elif bucketed:
                from .pipeline import PrefetchingLoader, preprocess_on_device
                from .decode_pool import default_workers
                avg = max(1, len(mine) // max(1, len(plan)))  # buckets are often smaller than batch_size
                depth = max(len(ring.streams) + 1, -(-2 * (workers or default_workers()) // avg))
                loader = PrefetchingLoader(dataset, mine, batch_size, workers=workers, depth=depth, pack=True,
                                           pack_size=None, batches=[[mine[p] for p in pos] for pos in plan])
                for pos, ((packed, shapes), gts, _) in zip(plan, loader):
                    yield pos, (shapes, lambda sh, S, dev, packed=packed, pad=padded(shapes), **kw:
                                preprocess_on_device(sh, None, dev, packed=packed, pad_to=pad, **kw)), gts
            elif input_pipeline == "host":
                for s in range(0, len(mine), batch_size):
                    items = [dataset[i] for i in mine[s:s + batch_size]]
                    yield s, torch.stack([it["x"] for it in items]), [it["m"].squeeze() for it in items]
            else:
                from .pipeline import PrefetchingLoader, preprocess_on_device
                s = 0
                # decode on the worker threads, packing into page-locked staging on one more thread, a batch ahead
                # enough batches in flight to keep every decode worker busy (batch 1: one image per batch)
                from .decode_pool import default_workers
                depth = max(len(ring.streams) + 1, -(-2 * (workers or default_workers()) // batch_size))
                for (packed, shapes), gts, _ in PrefetchingLoader(dataset, mine, batch_size, workers=workers,
                                                                 depth=depth, pack=True, pack_size=img_size):
                    yield s, (shapes, lambda sh, S, dev, packed=packed, **kw: preprocess_on_device(sh, S, dev, packed=packed, **kw)), gts
                    s += len(shapes)

        bucket_pos, bucket_rows = [], []
        try:
            if pin_path is not None:  # inside the try: whatever fails below, the shared model gets its "auto" back
                self.model.attention_path = pin_path
            for s, x, gts in batches():
                with ring.next():
                    u8 = None
                    if isinstance(x, tuple):  # decoded uint8 images: resize / normalise on this batch's stream
                        rgbs, pre = x
                        if refine:
                            x, u8 = pre(rgbs, img_size, device, pinned=True, return_u8=True)
                        else:
                            x = pre(rgbs, img_size, device, pinned=True)
                            x = x if (img_size is not None or bucketed) else x[0]
                    out = self._forward({"x": x}, device=device)
                    mask_pred, obj = out["mask_pred"], out.get("objectness")
                    if mask_pred.dim() == 5:  # evaluator.pyc@L199-205: last decoder layer
                        mask_pred, obj = mask_pred[:, -1], obj[:, -1]
                    gtb = ops.GtBatch.from_packed(gts, device) if isinstance(gts, tuple) else ops.GtBatch(gts, device)
                    rows = ops.evaluate_masks(mask_pred, obj.squeeze(-1), gtb, scale=scale)
                    if bucketed:  # s = this bucket's positions in the rank's image list: scattered after the loop - an index
                        bucket_pos += s       # tensor built here is a pageable host-to-device copy, which waits for this
                        bucket_rows.append(rows)  # stream's work: it serialised the three streams (2.2 ms per batch)
                    else:
                        rows_local[s:s + gtb.B] = rows
                    if refine:
                        from .bilateral_solver import bilateral_solver_batch_device
                        target = ops.upsample_selected(mask_pred, rows, (img_size, img_size), "pick")
                        _, binary = bilateral_solver_batch_device(u8, target)
                        refined = ops.mask_u8_to_f32(binary).unsqueeze(1)  # one "query" per image; its objectness is moot
                        rows_refined[s:s + gtb.B] = ops.evaluate_masks(refined, rows[:, 0:1], gtb, scale=0.0)
            ring.join()
            if bucket_rows:
                rows_local[torch.as_tensor(bucket_pos, device=device)] = torch.cat(bucket_rows)
        finally:
            if prev_path == "auto":
                self.model.attention_path = prev_path
        self.graph_stats = {"captures": self._graphed.captures, "replays": self._graphed.replays,
                            "failed": self._graphed.failed}
        self._graphed = None
        if self.debug:
            n_total = len(mine) * comm.world_size
            mine = list(range(comm.rank, n_total, comm.world_size))
        rows = gather_rows(rows_local, mine, n_total, comm)
        results = average_rows(rows)
        if comm.rank == 0:
            os.makedirs(dir_ckpt, exist_ok=True)
            with open(os.path.join(dir_ckpt, f"metrics_{dataset_name}.txt"), "w") as f:  # evaluator.pyc@L275-293
                f.write(HEADER)
                f.write(",".join(str(results[k + sfx]) for sfx in ("", "_ub") for k in
                                 ("iou", "pixel_accuarcy", "f_score", "f_max", "f_mean", "mae", "s_measure")))
        self.last_rows = rows
        if refine:
            rr = gather_rows(rows_refined, mine, n_total, comm)
            ref = average_rows(rr)
            results.update({k + "_refined": ref[k] for k in KEYS})
            self.last_rows_refined = rr
            if comm.rank == 0:
                with open(os.path.join(dir_ckpt, f"metrics_{dataset_name}_refined.txt"), "w") as f:
                    f.write("iou,pixel_acc,f_score,f_max,f_mean,mae,s_measure\n")
                    f.write(",".join(str(ref[k]) for k in ("iou", "pixel_accuarcy", "f_score", "f_max", "f_mean", "mae", "s_measure")))
        return results


def main(argv=None):
    """CLI of evaluator.pyc@L312-373: --config --p_state_dict --dataset_name ... (yaml merged over the flags)."""
    import yaml
    from .maskformer import load_checkpoint
    from .misc import get_model, set_seeds
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--p_state_dict", type=str, required=True)
    ap.add_argument("--dataset_name", type=str, default="duts", choices=["dut_omron", "duts", "ecssd"])
    ap.add_argument("--use_gpu", type=bool, default=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dir_root", type=str, default="..")
    ap.add_argument("--gpu_id", type=int, default=0)
    ap.add_argument("--suffix", type=str, default="")
    ap.add_argument("--img_size", type=int, default=None, help="batched mode: resize inputs to S x S")
    ap.add_argument("--batch_size", type=int, default=1)
    args = ap.parse_args(argv)
    base = yaml.safe_load(open(args.config))
    vars(args).update(base)
    set_seeds(args.seed)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else args.gpu_id
    if world > 1:
        import torch.distributed as dist
        from .distributed import pin_rank_cores
        pin_rank_cores()  # this rank's block of host cores (decode workers inherit it)
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    model = get_model("maskformer", configs=args)
    load_checkpoint(model, args.p_state_dict)
    model = model.to(device).eval()
    ev = Evaluator(network=model, dir_dataset=args.dir_dataset)
    ev.device = device
    res = ev(args.dataset_name, dir_ckpt=os.path.join(args.dir_ckpt, "eval" + args.suffix), img_size=args.img_size,
             scale_factor=args.scale_factor, batch_size=args.batch_size, device=device)
    if int(os.environ.get("RANK", "0")) == 0:
        print(res)
    return res


if __name__ == "__main__":
    main()
