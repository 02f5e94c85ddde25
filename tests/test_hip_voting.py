"""Pseudo-mask voting on the device (csrc/voting.hip) against the oracle (whose filter is pinned by the reference's own
utils.misc.filter_masks through tests/golden/voting.npz): surviving set, IoU table, row sums, winner."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import voting_oracle as V  # noqa: E402  (checker only)
from selfmask_amd.voting import vote_mask  # noqa: E402
from vote_known_answers import KNOWN_ANSWERS  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden", "voting.npz")
FLAGS = {"long": (True, False), "both": (True, True), "none": (False, False)}


@pytest.mark.parametrize("tag", sorted(FLAGS))
def test_vote_matches_oracle_and_reference_filter(tag):
    g = np.load(GOLD)
    for i in range(int(g["n_cases"])):
        masks = torch.from_numpy(g[f"masks_{i}"])
        best_mask, best, new_to_prev = vote_mask(masks.to(DEV), *FLAGS[tag])
        assert [new_to_prev[k] for k in range(len(new_to_prev))] == g[f"kept_{i}_{tag}"].tolist()  # the real filter_masks
        ref_mask, ref_best, ref_map, table, ious = V.vote_mask(masks, *FLAGS[tag])
        assert new_to_prev == ref_map and best == ref_best and torch.equal(best_mask.cpu(), ref_mask)
        last = vote_mask.last
        kept = [new_to_prev[k] for k in range(len(new_to_prev))]
        dev_table = last["iou"].cpu()[kept][:, kept]
        assert torch.equal(dev_table, table)                                   # integer counts -> identical fp32 quotients
        assert torch.allclose(last["row_sums"].cpu()[kept], ious, rtol=0, atol=2e-6)


def test_27_candidates_like_the_generator():
    """3 feature types x (2 + 3 + 4) one-hot cluster maps (mask_generator.pyc@L136-200), 27 candidates of one image."""
    rng = np.random.Generator(np.random.PCG64(7))
    h, w = 231, 317
    yy, xx = np.mgrid[:h, :w]
    cands = []
    for f in range(3):
        for k in (2, 3, 4):
            centers = rng.uniform(0.2, 0.8, size=(k, 2)) * (h, w)
            lab = np.argmin(((yy[None] - centers[:, 0, None, None]) ** 2 + (xx[None] - centers[:, 1, None, None]) ** 2), 0)
            cands += [(lab == c) for c in range(k)]
    masks = torch.from_numpy(np.stack(cands).astype(np.uint8))
    assert masks.shape[0] == 27
    best_mask, best, new_to_prev = vote_mask(masks.to(DEV))
    ref_mask, ref_best, ref_map, _, _ = V.vote_mask(masks)
    assert new_to_prev == ref_map and best == ref_best and torch.equal(best_mask.cpu(), ref_mask)


def test_everything_filtered_votes_over_all_candidates():
    """utils/misc.py:311-314: nothing survives -> dt_masks unfiltered + identity map (golden case 3 pins it with the real
    filter_masks); all-empty candidates: every IoU is 0 / (0 + 1e-7) = 0, the first index wins."""
    masks = torch.zeros(3, 40, 40, dtype=torch.uint8)
    best_mask, best, new_to_prev = vote_mask(masks.to(DEV))
    assert best == 0 and new_to_prev == {0: 0, 1: 1, 2: 2} and not best_mask.any()
    ref_mask, ref_best, ref_map, _, _ = V.vote_mask(masks)
    assert ref_best == best and ref_map == new_to_prev


@pytest.mark.parametrize("name", sorted(KNOWN_ANSWERS))
def test_vote_known_answers(name):
    """Hand-computed answers (tests/vote_known_answers.py): a tie, a single survivor, the all-filtered fallback."""
    masks, flags, want_best, want_map, want_table = KNOWN_ANSWERS[name]()
    best_mask, best, new_to_prev = vote_mask(torch.from_numpy(masks).to(DEV), *flags)
    assert best == want_best and new_to_prev == want_map
    assert np.array_equal(best_mask.cpu().numpy(), masks[want_map[want_best]])
    kept = [new_to_prev[k] for k in range(len(new_to_prev))]
    assert np.array_equal(vote_mask.last["iou"].cpu().numpy()[kept][:, kept], np.asarray(want_table, np.float32))


@pytest.mark.parametrize("h,w", [(224, 224), (48, 50), (37, 53), (300, 400), (16, 16), (5, 7)])
def test_pack_shapes_bytes_and_batches(h, w):
    """The bitmap / box pass in both its forms (16-byte loads when every mask starts 16-byte aligned, byte loads otherwise), rows that
    end inside a lane's 16 pixels, any non-zero byte counting as set, single pixels in the corners, and the batch form."""
    from selfmask_amd.voting import vote_mask_batch
    rng = np.random.Generator(np.random.PCG64(h * 1000 + w))
    yy, xx = np.mgrid[:h, :w]
    batch = []
    for b in range(3):
        ms = []
        for _ in range(5):
            cy, cx, ry, rx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w, rng.uniform(0.15, 0.4) * h, rng.uniform(0.15, 0.4) * w
            ms.append(((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).astype(np.uint8) * rng.choice([1, 2, 128, 255]))
        corner = np.zeros((h, w), np.uint8)
        corner[[0, -1, 0, -1], [0, -1, -1, 0]] = 255
        ms += [corner, (rng.random((h, w)) < 0.5).astype(np.uint8), np.zeros((h, w), np.uint8)]
        batch.append(np.stack(ms))
    masks = torch.from_numpy(np.stack(batch))
    got = vote_mask_batch(masks.to(DEV), False, False)
    tables = vote_mask_batch.last["iou"].cpu()
    for b in range(3):
        binary = (masks[b] != 0).to(torch.uint8)
        ref_mask, ref_best, ref_map, table, _ = V.vote_mask(binary, False, False)
        assert got[b][1] == ref_best and got[b][2] == ref_map
        assert torch.equal((got[b][0].cpu() != 0).to(torch.uint8), ref_mask)
        kept = [ref_map[k] for k in range(len(ref_map))]
        assert torch.equal(tables[b][kept][:, kept], table)
        for flags in ((True, False), (True, True)):  # the filters read the boxes and areas
            one = vote_mask(masks[b].to(DEV), *flags)
            ref = V.vote_mask(binary, *flags)
            assert one[1] == ref[1] and one[2] == ref[2]


@pytest.mark.parametrize("h,w", [(224, 224), (300, 400), (37, 53), (1, 9), (5, 1)])
def test_run_length_codes_from_the_device(h, w):
    """sm_rle_runs_u8 + the host's differences against mask_generator.rle_encode (COCO's uncompressed form: column-major, zeros first):
    blobs, a mask starting with a set pixel, all zeros, all ones, any non-zero byte, and more runs than the buffer holds (a second pass on the device with a longer one)."""
    from selfmask_amd.mask_generator import rle_decode, rle_encode
    from selfmask_amd.voting import rle_runs_async
    rng = np.random.Generator(np.random.PCG64(h * 7 + w))
    yy, xx = np.mgrid[:h, :w]
    blob = ((((yy - h * 0.4) / max(1.0, h * 0.3)) ** 2 + ((xx - w * 0.55) / max(1.0, w * 0.25)) ** 2) <= 1).astype(np.uint8)
    first = blob.copy()
    first[0, 0] = 200
    masks = np.stack([blob, first, np.zeros((h, w), np.uint8), np.full((h, w), 3, np.uint8), (rng.random((h, w)) < 0.5).astype(np.uint8) * 255])
    for cap in (8192, 4):
        got = rle_runs_async(torch.from_numpy(masks).to(DEV), cap=cap).result()
        for b in range(len(masks)):
            want = rle_encode(masks[b])
            assert got[b] == want, (b, cap)
            assert np.array_equal(rle_decode(got[b]), (masks[b] != 0).astype(np.uint8))


def test_images_of_different_sizes_in_one_batch():
    """sm_vote_masks_sized_u8 / sm_rle_runs_u8 with per-image sizes: candidates of 3 images inside planes of 64 x 80 (what sharing a patch
    grid gives), rubbish outside every image's own H x W - each image's vote and run-length code are those of its cropped candidates alone."""
    from selfmask_amd.mask_generator import rle_encode
    from selfmask_amd.voting import rle_runs_async, vote_mask_batch_async
    rng = np.random.Generator(np.random.PCG64(11))
    Hp, Wp, sizes = 64, 80, [(64, 80), (49, 66), (60, 80)]
    yy, xx = np.mgrid[:Hp, :Wp]
    planes = []
    for h, w in sizes:
        ms = []
        for _ in range(6):
            cy, cx, ry, rx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w, rng.uniform(0.15, 0.45) * h, rng.uniform(0.15, 0.45) * w
            m = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).astype(np.uint8)
            m[h:] = rng.integers(0, 2, size=m[h:].shape)      # what the padding's tokens were clustered into: must not count
            m[:, w:] = rng.integers(0, 2, size=m[:, w:].shape)
            ms.append(m)
        ms.append(np.pad(np.ones((h, 3), np.uint8), ((0, Hp - h), (2, Wp - 5))))  # a full-height stripe of the IMAGE: "long" only by its own H
        planes.append(np.stack(ms))
    masks = torch.from_numpy(np.stack(planes))
    for flags in ((True, False), (True, True), (False, False)):
        pend = vote_mask_batch_async(masks.to(DEV), *flags, winners=True, sizes=sizes)
        got, codes = pend.result(), rle_runs_async(pend.winners, sizes=sizes).result()
        for b, (h, w) in enumerate(sizes):
            crop = masks[b, :, :h, :w].contiguous()
            ref_mask, ref_best, ref_map, table, _ = V.vote_mask(crop, *flags)
            assert got[b][1] == ref_best and got[b][2] == ref_map, (b, flags)
            assert torch.equal(got[b][0].cpu()[:h, :w], ref_mask)
            kept = [ref_map[k] for k in range(len(ref_map))]
            assert torch.equal(pend.iou.cpu()[b][kept][:, kept], table)
            assert codes[b] == rle_encode(ref_mask.numpy())
