"""ORACLE (test infrastructure, NOT product code): CPU restatement of the pseudo-mask VOTING step.

``filter_masks`` / ``mask_to_bbox`` follow /root/reference/utils/misc.py:269-314 line by line (numpy + torch-CPU).
``vote_mask`` follows datasets/mask_generator (bytecode only; SURVEY.md Appendix B, @L202-230): filter -> IoU table with eps
1e-7 -> row sums -> argsort(descending)[0].  Parity status: filter_masks PINNED by tests/golden/voting.npz (produced by the
real utils.misc.filter_masks, oracle/gen_golden.py --only voting); vote_mask itself restated from the disassembly."""
from typing import Dict, Tuple

import numpy as np
import torch


def mask_to_bbox(mask: np.ndarray) -> Dict[int, Tuple[int, int, int, int]]:
    """utils/misc.py:269-282"""
    out = {}
    if mask.ndim == 2:
        mask = mask[None]
    for i, m in enumerate(mask):
        ys, xs = np.where(m)
        if ys.size == 0:  # "a mask which does not predict anything"
            continue
        out[i] = (int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()))
    return out


def filter_masks(dt_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False):
    """utils/misc.py:285-314 -> (stacked surviving masks, new_index_to_prev_index)"""
    kept, new_to_prev = [], {}
    h, w = dt_masks.shape[-2:]
    for idx, (ymin, ymax, xmin, xmax) in mask_to_bbox(dt_masks.cpu().numpy()).items():
        if remove_long_masks:
            if ymin == 0 and ymax + 1 == h:
                continue
            elif xmin == 0 and xmax + 1 == w:
                continue
        if remove_small_large_masks:
            if dt_masks[idx].sum() < 0.05 * h * w:
                continue
            elif (xmax - xmin) * (ymax - ymin) > 0.95 * h * w:
                continue
        new_to_prev[len(kept)] = idx
        kept.append(dt_masks[idx])
    if not kept:  # misc.py:311-314: "rare case where all predictions are filtered" -> everything back, identity map
        return dt_masks, {i: i for i in range(len(dt_masks))}
    return torch.stack(kept, dim=0), new_to_prev


def vote_mask(batch_pred_masks: torch.Tensor, remove_long_masks: bool = True, remove_small_large_masks: bool = False):
    """mask_generator.pyc@L202-230 -> (best mask, best index among the survivors, new_index_to_prev_index, iou table, row sums)"""
    masks, new_to_prev = filter_masks(batch_pred_masks, remove_long_masks, remove_small_large_masks)
    mb = masks.to(torch.bool)
    inter = torch.logical_and(mb[:, None], mb[None]).sum(dim=(-1, -2))
    union = torch.logical_or(mb[:, None], mb[None]).sum(dim=(-1, -2))
    table = inter / (union + 1e-7)
    ious = table.sum(dim=1)
    best = int(torch.argsort(ious, descending=True)[0])
    return masks[best], best, new_to_prev, table, ious
