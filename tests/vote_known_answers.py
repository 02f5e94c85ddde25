"""Hand-computed known-answer cases for the pseudo-mask vote (mask_generator.pyc@L202-230 + utils/misc.py:285-314): the
reference holds no vector for ``vote_mask`` (bytecode only), so these pin the restatement and the kernel by arithmetic a
reader can check: iou[i][j] = |mi & mj| / (|mi | mj| + 1e-7) in fp32 (4 + 1e-7 rounds to 4, so a mask's IoU with itself
is exactly 1), score = row sum over the survivors, winner = first maximum."""
import numpy as np


def _blank(n, h=6, w=6):
    return np.zeros((n, h, w), np.uint8)


def tie():
    """m0 = 2x2 block, m1 = 2x4 block containing it, m2 = a disjoint 2x2 block; nothing touches the border.
    iou(0,1) = 4/8, iou(0,2) = iou(1,2) = 0 -> row sums 1.5, 1.5, 1.0: m0 and m1 tie, the first one wins."""
    m = _blank(3)
    m[0, 1:3, 1:3] = 1
    m[1, 1:3, 1:5] = 1
    m[2, 3:5, 1:3] = 1
    return m, (True, False), 0, {0: 0, 1: 1, 2: 2}, [[1, .5, 0], [.5, 1, 0], [0, 0, 1]]


def single_survivor():
    """a full-height strip (long), an empty mask (no bbox) and one blob: only the blob survives -> index 0 of the survivors."""
    m = _blank(3)
    m[0, :, 0:2] = 1
    m[2, 2:4, 2:5] = 1
    return m, (True, False), 0, {0: 2}, [[1]]


def all_filtered():
    """a full-height strip and an empty mask: remove_long_masks drops the strip, the empty one has no bbox -> the fallback hands
    back BOTH with the identity map; iou(strip, strip) = 1, everything involving the empty mask is 0 -> the strip wins."""
    m = _blank(2)
    m[0, :, 0:2] = 1
    return m, (True, False), 0, {0: 0, 1: 1}, [[1, 0], [0, 0]]


def small_large():
    """remove_small_large_masks on 10x10: area < 0.05*100 = 5 is dropped (a 2x2 block), (xmax-xmin)*(ymax-ymin) > 95 is
    dropped only by a box spanning 11 px - impossible here, so the 8x8 block (inside the border) stays with the 3x3 one.
    iou(3x3 at (1..3), 8x8 at (1..8)) = 9/64."""
    m = _blank(3, 10, 10)
    m[0, 1:3, 1:3] = 1
    m[1, 1:4, 1:4] = 1
    m[2, 1:9, 1:9] = 1
    q = np.float32(9) / (np.float32(64) + np.float32(1e-7))
    return m, (True, True), 0, {0: 1, 1: 2}, [[1, q], [q, 1]]


KNOWN_ANSWERS = {"tie": tie, "single_survivor": single_survivor, "all_filtered": all_filtered, "small_large": small_large}
