"""End-to-end Evaluator on the device against the CPU oracle pipeline (oracle forward -> oracle post-processing ->
oracle metrics -> AverageMeter): BASELINE.json configs[0] ("evaluator.py on 16 ECSSD images, batch 1, native
resolution") with the shipped patch size 8, the batched 224^2 mode, and image sharding with two virtual ranks."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import evaluator_oracle as E  # noqa: E402  (checker only)
from oracle import selfmask_oracle as O  # noqa: E402
from selfmask_amd import MaskFormer, synthetic_state_dict  # noqa: E402
from selfmask_amd import datasets as DS, distributed as D  # noqa: E402
from selfmask_amd.evaluator import Evaluator  # noqa: E402

DEV = torch.device("cuda:0")


def _model(patch, seed, style="calib"):
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True,
                   use_binary_classifier=True)
    sd = synthetic_state_dict(seed, style, patch_size=patch)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV), sd


def _oracle_pipeline(dataset, sd, patch, scale_factor):
    rows = []
    for i in range(len(dataset)):
        it = dataset[i]
        out = O.forward(it["x"][None], sd, patch)
        gt = it["m"].to(torch.int64)
        pm, q, ub, _ = E.postprocess(out["mask_pred"][0, -1], out["objectness"][0, -1, :, 0], gt,
                                     scale_factor=scale_factor)
        rows.append(np.concatenate([E.all_metrics(pm[q], gt), E.all_metrics(pm[ub], gt), [q, ub]]))
    return np.array(rows)


def _check(res, rows_hip, rows_ref):
    assert np.array_equal(rows_hip[:, 14:], rows_ref[:, 14:])  # same picked / upper-bound queries
    # a 1e-5 logit difference can move a handful of pixels across 0.5: per-image values agree to ~1e-3 at worst,
    # the dataset averages to 3 d.p. (the north-star gate)
    assert np.abs(rows_hip[:, :14] - rows_ref[:, :14]).max() <= 2e-3
    ref = D.average_rows(rows_ref.astype(np.float32))
    for k, v in ref.items():
        assert abs(res[k] - v) < 5e-4, (k, res[k], v)
    assert round(res["iou"], 3) == round(ref["iou"], 3) or abs(res["iou"] - ref["iou"]) < 1e-4


def test_reference_mode_16_ecssd_images_patch8(tmp_path):
    """configs[0]: batch 1, native resolution, ViT-S/8 (the shipped yaml), up-sample x4 + crop."""
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 16, seed=7, size_range=(88, 136))
    model, sd = _model(8, 21)
    ev = Evaluator(network=model, dir_dataset=str(tmp_path))
    ev.device = DEV
    res = ev("ecssd", dir_ckpt=str(tmp_path / "ckpt"), scale_factor=2, batch_size=1, device=DEV)
    ref_rows = _oracle_pipeline(DS.get_dataset(str(tmp_path), "ecssd"), sd, 8, scale_factor=4)
    _check(res, ev.last_rows, ref_rows)
    txt = open(tmp_path / "ckpt" / "metrics_ecssd.txt").read().split("\n")
    assert txt[0] + "\n" == D.HEADER and len(txt[1].split(",")) == 14
    assert set(res) == {k + s for k in D.KEYS for s in ("", "_ub")}


def test_batched_mode_224_patch16_and_two_virtual_ranks(tmp_path):
    DS.write_synthetic_dataset(str(tmp_path), "duts", 11, seed=9, size_range=(150, 260))
    model, sd = _model(16, 22)
    ev = Evaluator(network=model, dir_dataset=str(tmp_path))
    ev.device = DEV
    res = ev("duts", dir_ckpt=str(tmp_path / "ckpt"), img_size=224, batch_size=4, device=DEV)
    ds = DS.get_dataset(str(tmp_path), "duts", eval_img_size=224)
    _check(res, ev.last_rows, _oracle_pipeline(ds, sd, 16, scale_factor=None))
    single_rows = ev.last_rows.copy()

    # image sharding: two ranks evaluated in turn through an in-process communicator; gathered rows and averages
    # must be bit-identical to the single-rank run (no RCCL needed to test the logic; bench.py exercises RCCL)
    class FakeComm:
        def __init__(self, rank, world, store):
            self.rank, self.world_size, self.store = rank, world, store

        def all_gather(self, t):
            self.store[self.rank] = t.clone()
            if len(self.store) < self.world_size:
                raise StopIteration  # first rank: park until the peer has contributed
            return torch.stack([self.store[r] for r in range(self.world_size)])

    store = {}
    try:
        ev("duts", dir_ckpt=str(tmp_path / "c0"), img_size=224, batch_size=4, device=DEV, comm=FakeComm(0, 2, store))
    except StopIteration:
        pass
    res2 = ev("duts", dir_ckpt=str(tmp_path / "c1"), img_size=224, batch_size=4, device=DEV, comm=FakeComm(1, 2, store))
    assert np.array_equal(ev.last_rows, single_rows)
    assert res2 == res


def test_batches_in_flight_on_several_streams_give_identical_rows(tmp_path):
    """streams.StreamRing: 1, 2 and 3 batches in flight run the same kernels per batch -> bit-identical result rows
    (different batch shapes share the model: each (shape, stream) pair has its own workspace)."""
    DS.write_synthetic_dataset(str(tmp_path), "duts", 22, seed=10, size_range=(120, 230))
    model, _ = _model(16, 23)
    ev = Evaluator(network=model, dir_dataset=str(tmp_path))
    ev.device = DEV
    rows = []
    for n in (1, 2, 3):
        ev("duts", dir_ckpt=str(tmp_path / f"s{n}"), img_size=224, batch_size=3, device=DEV, streams=n)
        rows.append(ev.last_rows.copy())
        # 22 images in batches of 3 = 7 full batches + 1 ragged: a stream that sees the full shape three times captures it
        assert ev.graph_stats["failed"] is None and ev.graph_stats["replays"] >= 1
    assert np.array_equal(rows[0], rows[1]) and np.array_equal(rows[0], rows[2])
    ev("duts", dir_ckpt=str(tmp_path / "eager"), img_size=224, batch_size=3, device=DEV, streams=2, hip_graph=False)
    assert np.array_equal(rows[0], ev.last_rows) and ev.graph_stats["replays"] == 0


def test_reference_mode_16_ecssd_images_patch16(tmp_path):
    """configs[0] as BASELINE.json words it: ViT-S/16, nq = 20, batch 1, native resolution (the evaluator's hard-coded x4
    generalised to patch_size // scale_factor = 8)."""
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 16, seed=17, size_range=(120, 200))
    model, sd = _model(16, 24)
    ev = Evaluator(network=model, dir_dataset=str(tmp_path))
    ev.device = DEV
    res = ev("ecssd", dir_ckpt=str(tmp_path / "ckpt"), scale_factor=2, batch_size=1, device=DEV)
    ref_rows = _oracle_pipeline(DS.get_dataset(str(tmp_path), "ecssd"), sd, 16, scale_factor=8)
    _check(res, ev.last_rows, ref_rows)
    assert ev.graph_stats["captures"] == 0  # native resolution: a new shape per image, graphs stay off


def test_configs2_384_bilateral_refinement_end_to_end(tmp_path):
    """BASELINE.json configs[2]: 384^2 inputs, picked mask -> x8 up-sample -> bilateral solver (the whole batch in one
    launch sequence) -> metrics of the refined binary mask, against the oracle chained the same way (oracle forward ->
    post-processing -> F.interpolate x8 -> numpy solver restatement -> metrics)."""
    from oracle import bilateral_oracle as BO
    from PIL import Image
    import torch.nn.functional as F
    S, n_img = 384, 3
    DS.write_synthetic_dataset(str(tmp_path), "duts", n_img, seed=31, size_range=(300, 400))
    model, sd = _model(16, 26)
    ev = Evaluator(network=model, dir_dataset=str(tmp_path))
    ev.device = DEV
    res = ev("duts", dir_ckpt=str(tmp_path / "ckpt"), img_size=S, batch_size=n_img, device=DEV, refine="bilateral", workers=2)
    ds = DS.get_dataset(str(tmp_path), "duts", eval_img_size=S)
    rows_ref, rows_refined_ref, flips = [], [], []
    for i in range(n_img):
        it = ds[i]
        out = O.forward(it["x"][None], sd, 16)
        gt = it["m"].to(torch.int64)
        pm, q, ub, _ = E.postprocess(out["mask_pred"][0, -1], out["objectness"][0, -1, :, 0], gt, scale_factor=None)
        rows_ref.append(np.concatenate([E.all_metrics(pm[q], gt), E.all_metrics(pm[ub], gt), [q, ub]]))
        target = F.interpolate(out["mask_pred"][0, -1, q][None, None], size=(S, S), mode="bilinear", align_corners=False)[0, 0]
        rgb = np.asarray(Image.open(ds.p_imgs[i]).convert("RGB").resize((S, S), Image.BILINEAR))
        soft, binary = BO.bilateral_solver_output(rgb, target.double().numpy())[:2]
        refined = F.interpolate(torch.from_numpy(binary.astype(np.float32))[None, None], size=tuple(gt.shape), mode="bilinear",
                                align_corners=False)[0, 0]
        rows_refined_ref.append(E.all_metrics(refined, gt))
    _check({k: v for k, v in res.items() if not k.endswith("_refined")}, ev.last_rows, np.array(rows_ref))
    got = ev.last_rows_refined[:, :7]
    ref = np.array(rows_refined_ref)
    print("\nconfigs[2] refined metrics (hip | oracle):\n", np.round(got, 5), "\n", np.round(ref, 5))
    assert np.abs(got - ref).max() <= 2e-3
    assert set(res) == {k + s for k in D.KEYS for s in ("", "_ub", "_refined")}
    assert os.path.exists(tmp_path / "ckpt" / "metrics_duts_refined.txt")


@pytest.mark.parametrize("patch,pipe", [(16, "device"), (8, "device"), (16, "host")])
def test_native_resolution_buckets_equal_batch1_rows_bit_for_bit(tmp_path, patch, pipe):
    """Native-resolution evaluation batched by token grid (batch_size > 1 with img_size=None): each image zero-padded into
    its bucket's (B, 3, gh P, gw P) batch.  Every result row must equal the reference's own mode (batch 1) bit for bit."""
    DS.write_synthetic_dataset(str(tmp_path), "ecssd", 26, seed=5, size_range=(90, 150))
    model, _ = _model(patch, 23)
    ev = Evaluator(network=model, dir_dataset=str(tmp_path))
    ev.device = DEV
    ev("ecssd", dir_ckpt=str(tmp_path / "ckpt1"), batch_size=1, device=DEV, input_pipeline=pipe)
    rows1 = ev.last_rows.copy()
    res = ev("ecssd", dir_ckpt=str(tmp_path / "ckpt8"), batch_size=8, device=DEV, input_pipeline=pipe)
    assert np.array_equal(ev.last_rows, rows1)
    ref = D.average_rows(rows1)
    assert all(res[k] == ref[k] for k in ref)
    # and through graph replay (bucket shapes are captured from their 32nd sighting per stream on by default: lowered here)
    ev.bucket_graph_admit_after = 0
    ev("ecssd", dir_ckpt=str(tmp_path / "ckpt8"), batch_size=2, device=DEV, input_pipeline=pipe, streams=1)
    assert np.array_equal(ev.last_rows, rows1)
    if pipe == "device":
        assert ev.graph_stats["failed"] is None and ev.graph_stats["captures"] >= 1
