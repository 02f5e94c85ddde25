"""The metrics restatement (oracle/evaluator_oracle.py) against known-answer vectors from the reference's metrics/*.py."""
import os

import numpy as np
import torch

from oracle import evaluator_oracle as E

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics.npz")
# golden column order (oracle/gen_golden.py): iou, f_score, f_max, f_mean, mae, pixel_acc, s_measure
GOLD_TO_ORACLE = [0, 5, 1, 2, 3, 4, 6]


def test_metrics_match_reference_bit_for_bit():
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        pred, gt = torch.from_numpy(g[f"pred_{i}"]), torch.from_numpy(g[f"gt_{i}"].astype(np.int64))
        ours = E.all_metrics(pred, gt)
        ref = g[f"vals_{i}"][GOLD_TO_ORACLE]
        assert np.array_equal(ours, ref), (i, ours, ref)


def test_average_meter_sequential_fp32():
    g = np.load(GOLD)
    meters = [E.AverageMeter() for _ in range(7)]
    for i in range(int(g["n_cases"])):
        pred, gt = torch.from_numpy(g[f"pred_{i}"]), torch.from_numpy(g[f"gt_{i}"].astype(np.int64))
        f = E.FMeasure()(pred, gt)
        vals = [E.compute_iou(pred, gt).numpy(), f["f_measure"].numpy(), f["f_max"].numpy(), f["f_mean"].numpy(),
                E.compute_mae(pred, gt).numpy(), E.compute_pixel_accuracy(pred, gt).numpy(),
                E.s_measure(pred, gt.to(torch.float32))]
        for m, v in zip(meters, vals):
            m.update(val=v, n=1)
    assert np.array_equal(np.array([np.float64(m.avg) for m in meters]), g["avg"])
