"""What the bytecode-only callers of the reference pin STATICALLY: `evaluator.py` and `datasets/mask_generator.py`
exist only as CPython 3.9 / 3.12 .pyc, which the image's 3.10 cannot import.  oracle/pyc_constants.py reads the 3.9 files as data
(marshal, nothing executed) into tests/golden/evaluator_constants.json; the product's strings, keys, keyword sets, thresholds and
defaults are asserted against it here.  The ORCHESTRATION around these constants stays 'restated from disassembly' (parity unpinned)."""
import json
import os

import pytest

from selfmask_amd import distributed as D
from selfmask_amd import evaluator as E
from selfmask_amd import voting as VT

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pins(golden_dir):
    with open(os.path.join(golden_dir, "evaluator_constants.json")) as f:
        return json.load(f)


def fn(pins, mod, name):
    return pins[mod]["functions"][name]


def test_evaluator_header_keys_and_kwargs(pins):
    call = fn(pins, "evaluator", "<module>.Evaluator.__call__")
    c = call["consts"]
    assert D.HEADER in c  # the metrics_<dataset>.txt header, byte for byte (note `miou_ub`)
    keys14 = [k + s for s in ("", "_ub") for k in D.KEYS]
    assert keys14 in c  # the returned dict's keys in order, `pixel_accuarcy` spelling included
    assert call["argnames"] == ["self", "dataset_name", "dir_ckpt", "img_size", "scale_factor", "batch_size", "device", "cost_type"]
    # F.interpolate(mask_pred, scale_factor=4, mode="bilinear", align_corners=False)
    assert list(E.REFERENCE_UPSAMPLE) in c
    for v in E.REFERENCE_UPSAMPLE.values():
        assert v in c
    assert "interpolate" in call["names"]
    assert E.MASK_THRESHOLD in c and ["descending"] in c and True in c and "argsort" in call["names"]
    assert E.VISUALISE_EVERY in c and ["shuffle", "batch_size", "num_workers", "with_tbar"] in c and E.DATALOADER_WORKERS in c
    assert "/metrics_" in c and ".txt" in c and "," in c
    assert "use_binary_classifier" in call["names"] and "objectness" in c and "mask_pred" in c
    # defaults of __call__ and __init__ live in the class body's constants
    body = fn(pins, "evaluator", "<module>.Evaluator")["consts"]
    assert ["vit_small", "/scratch/shared/beegfs/gyungin/datasets", None, False] in body and ["iou"] in body and 2 in body and 1 in body
    init = fn(pins, "evaluator", "<module>.Evaluator.__init__")
    assert init["argnames"] == ["self", "network", "arch", "dir_dataset", "visualizer", "debug"] and " does not exist." in init["consts"]
    meters = fn(pins, "evaluator", "<module>.Evaluator._init_meters")["names"]
    for k in ("f_score", "f_max", "f_mean", "mae", "iou", "pixel_acc", "s_measure"):
        assert k in meters and k + "_ub" in meters
    upd = fn(pins, "evaluator", "<module>.Evaluator._update_meters")
    assert ["val", "n"] in upd["consts"] and "f_measure" in upd["consts"] and "float32" in upd["names"]
    ub = fn(pins, "evaluator", "<module>.Evaluator._get_upper_bound_mask")
    assert "argmax" in ub["names"] and "argmin" in ub["names"] and ["f_measure", "f_max"] in ub["consts"]
    main = fn(pins, "evaluator", "<module>")["consts"]
    assert ["dut_omron", "duts", "ecssd"] in main and ["dataset_name", "dir_ckpt", "scale_factor", "batch_size"] in main


def test_evaluator_cli_mirror_has_the_reference_flags(pins):
    main = fn(pins, "evaluator", "<module>")["consts"]
    src = open(os.path.join(REPO, "salient-object-detection_amd", "selfmask_amd", "evaluator.py")).read()
    for flag in ("--config", "--p_state_dict", "--dataset_name", "--use_gpu", "--seed", "--dir_root", "--gpu_id", "--suffix"):
        assert flag in main and f'"{flag}"' in src


def test_mask_generator_defaults_thresholds_and_kwargs(pins):
    body = fn(pins, "mask_generator", "<module>.MaskGenerator")["consts"]
    assert list(VT.DEFAULT_CLUSTER_SIZES) in body and VT.DEFAULT_CLUSTER_TYPE in body
    assert [True, False] in body  # vote_mask(remove_long_masks=True, remove_small_large_masks=False)
    import inspect
    sig = inspect.signature(VT.vote_mask).parameters
    assert sig["remove_long_masks"].default is True and sig["remove_small_large_masks"].default is False
    sig = inspect.signature(VT.extract_candidate_masks).parameters
    assert sig["cluster_sizes"].default == VT.DEFAULT_CLUSTER_SIZES and sig["cluster_type"].default == VT.DEFAULT_CLUSTER_TYPE
    init = fn(pins, "mask_generator", "<module>.MaskGenerator.__init__")
    assert list(VT.CLUSTER_TYPES) in init["consts"] and "KMeansClustering" in init["names"] and "SpectralClustering" in init["names"]
    ext = fn(pins, "mask_generator", "<module>.MaskGenerator.extract_candidate_masks")["consts"]
    assert list(VT.FEATURE_UPSAMPLE) in ext and all(v in ext for v in VT.FEATURE_UPSAMPLE.values())
    assert VT.MASK_UPSAMPLE_MODE in ext and ["scale_factor", "mode"] in ext and VT.DINO_TOTAL_STRIDE in ext and 8 in ext
    assert "layer12" in ext and ["arch", "training_method", "patch_size"] in ext
    flt = fn(pins, "mask_generator", "<module>.MaskGenerator.filter_masks")["consts"]
    vote = fn(pins, "mask_generator", "<module>.MaskGenerator.vote_mask")
    assert 0.05 in flt and 0.95 in flt and 1e-07 in vote["consts"] and ["descending"] in vote["consts"]
    hip = open(os.path.join(REPO, "salient-object-detection_amd", "csrc", "voting.hip")).read()
    assert "0.05 * H * W" in hip and "0.95 * H * W" in hip and "1e-7f" in hip  # the literals the kernels compile in
    main = fn(pins, "mask_generator", "<module>")["consts"]
    assert "spectral" in main and "k-means" in main and [2, 3, 4] in main and 16 in main


def _in_order(names, *wanted):
    pos = [names.index(w) for w in wanted]
    return pos == sorted(pos)


def test_first_use_order_of_the_callers_names(pins):
    """`co_names` lists a function's global / attribute names in order of FIRST USE: a static trace of the order of work.  The mirror's
    order of work (evaluator.py, voting.py, mask_generator.py) is asserted against it."""
    call = fn(pins, "evaluator", "<module>.Evaluator.__call__")["names"]
    assert _in_order(call, "_init_meters", "get_dataset", "get_dataloader", "next", "_forward", "interpolate", "_get_upper_bound_mask",
                     "argsort", "_update_meters", "set_description", "makedirs", "einsum", "_visualize", "open", "write", "close")
    upd = fn(pins, "evaluator", "<module>.Evaluator._update_meters")["names"]
    # metrics computed first, meters updated in the order iou, f_score, f_max, f_mean, s_measure, mae, pixel_acc - then the same for *_ub
    assert _in_order(upd, "compute_iou", "FMeasure", "compute_mae", "compute_pixel_accuracy", "iou", "update", "f_score", "f_max", "f_mean",
                     "s_measure", "SMeasure", "mae", "pixel_acc", "iou_ub", "f_score_ub", "f_max_ub", "f_mean_ub", "s_measure_ub", "mae_ub",
                     "pixel_acc_ub")
    ext = fn(pins, "mask_generator", "<module>.MaskGenerator.extract_candidate_masks")["names"]
    assert _in_order(ext, "feature_types", "get_model", "DataLoader", "CustomDataset", "pad_input_image", "encoder", "interpolate",
                     "cluster_sizes", "clusterer", "to_one_hot", "concatenate")
    vote = fn(pins, "mask_generator", "<module>.MaskGenerator.vote_mask")["names"]
    assert _in_order(vote, "mask_to_bbox", "filter_masks", "logical_and", "sum", "logical_or", "argsort", "item")
    gen = fn(pins, "mask_generator", "<module>.MaskGenerator.__call__")["names"]
    assert _in_order(gen, "extract_candidate_masks", "vote_mask", "encode", "asfortranarray")
    # the written line follows the header's column order: the avg values of the seven meters, then the seven *_ub ones
    tail = call[call.index("f_score"):]
    for k in ("f_score", "f_max", "mae", "s_measure", "iou", "pixel_acc"):
        assert k in tail and k + "_ub" in tail
