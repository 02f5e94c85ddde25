"""Image-sharded evaluation: rank r takes images r, r+W, ...; ONE all-gather of per-image result rows; every rank
then reduces in global index order with the reference's sequential AverageMeter arithmetic, so the W-rank result is
bit-identical to the 1-rank result.  The reference has no distributed code at all."""
from typing import List, Optional, Sequence

import numpy as np
import torch

ROW = 16  # 14 metric values + picked query + upper-bound query
HEADER = ("iou,pixel_acc,f_score,f_max,f_mean,mae,s_measure,miou_ub,pixel_acc_ub,f_score_ub,f_max_ub,f_mean_ub,"
          "mae_ub,s_measure_ub\n")  # evaluator.pyc@L276 (note miou_ub vs the dict key iou_ub)
KEYS = ("iou", "pixel_accuarcy", "f_score", "f_max", "f_mean", "mae", "s_measure")  # dict keys, evaluator.pyc@L294-308


def rank_cores(local_rank: int, local_world: int, cores: Sequence[int]) -> List[int]:
    """This rank's share of the node's cores: a contiguous block (neighbouring cores share caches / a NUMA node)."""
    cores = sorted(cores)
    per = max(1, len(cores) // max(1, local_world))
    lo = min(local_rank * per, max(0, len(cores) - per))
    return cores[lo:lo + per]


def pin_rank_cores() -> List[int]:
    """One process per GPU, eight of them on one host: give each rank its own block of the host's cores (its decode workers
    inherit it) instead of 8 x N threads fighting over all of them.  Reads LOCAL_RANK / LOCAL_WORLD_SIZE (torch.distributed.run);
    a no-op for a single rank.  Sets SM_RANK_CORES_PINNED so that decode_pool.default_workers does not divide again."""
    import os
    lw, lr = int(os.environ.get("LOCAL_WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    try:
        cur = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    if lw <= 1 or os.environ.get("SM_RANK_CORES_PINNED") == "1":
        return cur
    mine = rank_cores(lr, lw, cur)
    os.sched_setaffinity(0, mine)
    os.environ["SM_RANK_CORES_PINNED"] = "1"
    return mine


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    return list(range(rank, n_items, world_size))


class TorchDistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, or "gloo" on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        # concatenated along dim 0 (the layout both the nccl/RCCL and the gloo backends accept), viewed as stacked
        src = t.contiguous()
        if src.is_cuda and self.dist.get_backend(self.group) == "gloo":  # gloo gathers host tensors (CPU tests, one-GPU rehearsals)
            src = src.cpu()
        out = torch.empty((self.world_size * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        self.dist.all_gather_into_tensor(out, src, group=self.group)
        return out.view((self.world_size,) + tuple(t.shape)).to(t.device)


class SingleComm:
    rank, world_size = 0, 1

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        return t.unsqueeze(0)


def gather_rows(rows_local: torch.Tensor, idx_local: Sequence[int], n_total: int, comm) -> np.ndarray:
    """rows_local (n_local, 16) on any device -> (n_total, 16) float32 numpy in global index order on every rank.
    Payload per rank: ceil(n/W) x 17 fp32 (global index + row), padded with index -1."""
    per = -(-n_total // comm.world_size)
    buf = torch.full((per, ROW + 1), -1.0, dtype=torch.float32, device=rows_local.device)
    n = len(idx_local)
    if n:
        buf[:n, 0] = torch.tensor(list(idx_local), dtype=torch.float32, device=rows_local.device)
        buf[:n, 1:] = rows_local
    allb = comm.all_gather(buf).reshape(-1, ROW + 1).cpu().numpy()
    out = np.full((n_total, ROW), np.nan, np.float32)
    seen = np.zeros(n_total, bool)
    for r in allb:
        i = int(r[0])
        if i >= 0:
            assert not seen[i], f"image {i} evaluated twice"
            out[i], seen[i] = r[1:], True
    assert seen.all(), f"{int((~seen).sum())} images missing after the gather"
    return out


def average_rows(rows: np.ndarray) -> dict:
    """metrics/average_meter.py:12-16 over images in index order: float32 running sums for the tensor-derived
    values, Python-float (fp64) sums for the S-measure (SMeasure returns ``Q.item()``), evaluator.pyc@L55-99."""
    out = {}
    for k in range(14):
        col = rows[:, k]
        if k % 7 == 6:  # s_measure / s_measure_ub
            s = 0.0
            for v in col:
                s += float(v)
            avg = s / len(col)
        else:
            s = np.float32(0)
            for v in col:
                s = np.float32(s + np.float32(v))
            avg = float(np.float32(s / np.float32(len(col))))
        out[KEYS[k % 7] + ("_ub" if k >= 7 else "")] = avg
    return out
