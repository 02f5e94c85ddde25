"""Mirror of the voting step of the reference's pseudo-mask generator: ``utils.misc.filter_masks`` (utils/misc.py:285-314)
and ``MaskGenerator.vote_mask`` (datasets/mask_generator, bytecode only: SURVEY.md Appendix B @L202-230) on the MI355X.

Only the voting is here.  The clustering that produces the candidate masks (``clusterings.SpectralClustering``: faiss k-NN
affinity + eigen-decomposition) does not exist in the reference in any form and is out of scope."""
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
