"""ORACLE (test infrastructure, NOT product code): CPU restatement of the evaluator post-processing and the metrics.

Follows /root/reference/metrics/*.py line by line (torch-CPU ops, same dtypes and operation order) and the
post-processing recovered from the evaluator bytecode.  Parity status: PINNED for the metrics
by tests/golden/metrics.npz (values produced by the real reference modules, oracle/gen_golden.py); the evaluator
itself exists in the reference only as CPython-3.9/3.12 bytecode that this interpreter cannot import, so its
post-processing is pinned only through the stock torch ops it calls (F.interpolate, argsort) - "restated from
disassembly", SURVEY.md section 8c.
"""
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

METRIC_NAMES = ("iou", "pixel_acc", "f_score", "f_max", "f_mean", "mae", "s_measure")  # header order, evaluator.pyc@L276


def compute_iou(pred_mask: torch.Tensor, gt_mask: torch.Tensor, threshold: Optional[float] = 0.5, eps: float = 1e-7):
    """metrics/iou.py:6-32"""
    if threshold is not None:
        pred_mask = pred_mask > threshold
    inter = torch.logical_and(pred_mask, gt_mask).sum(dim=(-1, -2))
    union = torch.logical_or(pred_mask, gt_mask).sum(dim=(-1, -2))
    return inter / (union + eps)


def compute_mae(pred_mask, gt_mask):
    """metrics/mae.py:4-9"""
    return torch.mean(torch.abs(pred_mask - gt_mask.to(torch.float32)), dim=(-1, -2))


def compute_pixel_accuracy(pred_mask, gt_mask, threshold=0.5):
    """metrics/pixel_acc.py:5-14"""
    return ((pred_mask > threshold) == gt_mask).to(torch.float32).mean(dim=(-1, -2))


class FMeasure:
    """metrics/f_measure.py:4-92 (beta_square is squared AGAIN in the formula: 0.3**2 = 0.09, a reference quirk)."""

    def __init__(self, default_thres=0.5, beta_square=0.3, n_bins=255, eps=1e-7):
        self.beta_square, self.default_thres, self.eps, self.n_bins = beta_square, default_thres, eps, n_bins

    def _pr(self, b, g):
        tp = torch.logical_and(b, g).sum(dim=(-1, -2))
        return tp / (b.sum(dim=(-1, -2)) + self.eps), tp / (g.sum(dim=(-1, -2)) + self.eps)

    def _f(self, prec, recall):
        b2 = self.beta_square ** 2
        return ((1 + b2) * prec * recall) / (b2 * prec + recall + self.eps)

    def __call__(self, pred, gt) -> Dict[str, torch.Tensor]:
        out = {"f_measure": self._f(*self._pr(pred > self.default_thres, gt))}
        thr = torch.arange(0, 1, 1 / self.n_bins).view(self.n_bins, 1, 1)
        out["f_max"] = torch.max(self._f(*self._pr(pred.unsqueeze(0).repeat(self.n_bins, 1, 1) > thr,
                                                   gt.unsqueeze(0).repeat(self.n_bins, 1, 1))))
        out["f_mean"] = self._f(*self._pr(pred > 2 * pred.mean(dim=(-1, -2), keepdim=True), gt))
        return out


def s_measure(pred: torch.Tensor, gt: torch.Tensor, alpha: float = 0.5) -> float:
    """metrics/s_measure.py:6-124 (CPU branch; gt float32 {0,1})."""
    gt = gt.clone()

    def ssim(p, g):
        g = g.float()
        h, w = p.shape[-2:]
        n = h * w
        x, y = p.mean(), g.mean()
        sx2 = ((p - x) * (p - x)).sum() / (n - 1 + 1e-20)
        sy2 = ((g - y) * (g - y)).sum() / (n - 1 + 1e-20)
        sxy = ((p - x) * (g - y)).sum() / (n - 1 + 1e-20)
        a = 4 * x * y * sxy
        b = (x * x + y * y) * (sx2 + sy2)
        if a != 0:
            return a / (b + 1e-20)
        if a == 0 and b == 0:
            return 1.0
        return 0

    def obj(p, g):
        t = p[g == 1]
        x, s = t.mean(), t.std()
        return 2.0 * x / (x * x + 1.0 + s + 1e-20)

    y = gt.mean()
    if y == 0:
        q = 1.0 - pred.mean()
    elif y == 1:
        q = pred.mean()
    else:
        gt[gt >= 0.5] = 1
        gt[gt < 0.5] = 0
        fg = torch.where(gt == 0, torch.zeros_like(pred), pred)
        bg = torch.where(gt == 1, torch.zeros_like(pred), 1 - pred)
        u = gt.mean()
        s_obj = u * obj(fg, gt) + (1 - u) * obj(bg, 1 - gt)
        rows, cols = gt.shape[-2:]
        total = gt.sum()
        i = torch.from_numpy(np.arange(0, cols)).float()
        j = torch.from_numpy(np.arange(0, rows)).float()
        X = torch.round((gt.sum(dim=0) * i).sum() / total).long()
        Y = torch.round((gt.sum(dim=1) * j).sum() / total).long()
        area = rows * cols
        w1 = X.float() * Y.float() / area
        w2 = (cols - X.float()) * Y.float() / area
        w3 = X.float() * (rows - Y.float()) / area
        w4 = 1 - w1 - w2 - w3
        s_reg = (w1 * ssim(pred[:Y, :X], gt[:Y, :X]) + w2 * ssim(pred[:Y, X:], gt[:Y, X:]) +
                 w3 * ssim(pred[Y:, :X], gt[Y:, :X]) + w4 * ssim(pred[Y:, X:], gt[Y:, X:]))
        q = alpha * s_obj + (1 - alpha) * s_reg
        if q.item() < 0:
            q = torch.FloatTensor([0.0])
    return q.item() if torch.is_tensor(q) else float(q)


def all_metrics(pred: torch.Tensor, gt: torch.Tensor) -> np.ndarray:
    """The seven per-image values in the order of evaluator.pyc@L276's header (iou, pixel_acc, f_score, f_max,
    f_mean, mae, s_measure).  pred (H,W) float32 in [0,1]; gt (H,W) integer {0,1}."""
    f = FMeasure()(pred, gt)
    return np.array([
        np.float64(compute_iou(pred, gt).numpy()), np.float64(compute_pixel_accuracy(pred, gt).numpy()),
        np.float64(f["f_measure"].numpy()), np.float64(f["f_max"].numpy()), np.float64(f["f_mean"].numpy()),
        np.float64(compute_mae(pred, gt).numpy()), np.float64(s_measure(pred, gt.to(torch.float32)))])


def postprocess(mask_pred_last: torch.Tensor, objectness_last: torch.Tensor, gt: torch.Tensor,
                scale_factor: Optional[int] = None):
    """evaluator.pyc@L199-228 for ONE image.  mask_pred_last (nq, h', w') probabilities, objectness_last (nq,),
    gt (H, W) {0,1}.  Reference mode (``scale_factor`` given, evaluator.pyc@L209-211: 4 for ViT-S/8; generalised to
    patch_size // 2): F.interpolate(scale_factor, bilinear, align_corners=False)[..., :H, :W].  Batched mode
    (``scale_factor=None``): F.interpolate(size=(H, W)).
    Returns (pred_masks (nq,H,W), q_star, ub, ious (nq,))."""
    H, W = gt.shape[-2:]
    pm = mask_pred_last.unsqueeze(0)
    if scale_factor is not None:
        pm = F.interpolate(pm, scale_factor=scale_factor, mode="bilinear", align_corners=False)[..., :H, :W]
    else:
        pm = F.interpolate(pm, size=(H, W), mode="bilinear", align_corners=False)
    pm = pm[0]
    ious = compute_iou(pm > 0.5, gt.unsqueeze(0).repeat(pm.shape[0], 1, 1))  # bool > 0.5 is a no-op (@L101-134)
    ub = int(torch.argmax(ious))
    q_star = int(torch.argsort(objectness_last, descending=True)[0])
    return pm, q_star, ub, ious


class AverageMeter:
    """metrics/average_meter.py:1-16 - sequential running sums (numpy float32 for tensor-derived values)."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n: int):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
