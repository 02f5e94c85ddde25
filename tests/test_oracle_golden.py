"""The CPU restatement (oracle/selfmask_oracle.py) against vectors produced by the REAL reference
(oracle/gen_golden.py, run in the build container).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import selfmask_oracle as O
from selfmask_amd.state_layout import synthetic_state_dict, synthetic_images, state_shapes

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "forward_*.npz")))


def test_state_layout_counts():
    s = state_shapes(20, 16, 6, True)
    assert len(s) == 267
    assert sum(int(np.prod(v)) for v in s.values()) == 36169729  # SURVEY.md section 8b probe
    assert state_shapes(20, 8)["encoder.pos_embed"] == (1, 785, 384)
    assert state_shapes(20, 16)["encoder.pos_embed"] == (1, 197, 384)


@pytest.mark.parametrize("fp", CASES, ids=[os.path.basename(c)[8:-4] for c in CASES])
def test_oracle_matches_reference_vectors(fp):
    g = np.load(fp)
    patch, B, Hh, Ww, wseed, xseed, nthreads = [int(v) for v in g["meta"]]
    style = str(g["style"])
    torch.set_num_threads(min(nthreads, os.cpu_count() or 1))
    sd = synthetic_state_dict(wseed, style, patch_size=patch)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    out = O.forward(x, sd, patch)
    scale = float(g["logit_absmax"])
    # same torch ops in the same order: bit-identical for the B=1 cases at equal thread count, and within
    # 1.3e-6 of the logit scale for B=2 (torch picks different bmm/upsample paths for the reference's strided
    # views).  Allow fp32 re-blocking noise of a different core count: 2e-6 relative to the logit scale.
    tol = 2e-6 * scale + 1e-6
    d = np.abs(out["mask_logits"][:, -1].numpy() - g["logits_last"]).max()
    assert d <= tol, (d, tol)
    assert np.abs(out["objectness"].numpy() - g["objectness"]).max() <= 2e-6
    assert np.abs(out["features"].numpy() - g["features"]).max() <= 2e-5
    assert np.abs(out["queries"].numpy() - g["queries"]).max() <= 2e-5
    assert tuple(g["grid"]) == (out["mask_logits"].shape[-2] // 2, out["mask_logits"].shape[-1] // 2)
    if "logits_all" in g:
        assert np.abs(out["mask_logits"].numpy() - g["logits_all"]).max() <= tol
        assert np.abs(out["patch_tokens"][0].numpy() - g["patch_tokens_b0"]).max() <= 2e-5
        assert np.abs(out["mask_pred"][:, -1].numpy() - g["mask_pred_last"]).max() <= 0.25 * tol + 1e-6


def test_oracle_fp64_mode_matches_reference_fp64():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "forward_p16_224_calib.npz"))
    patch, B, Hh, Ww, wseed, xseed, _ = [int(v) for v in g["meta"]]
    sd = O.cast_state(synthetic_state_dict(wseed, "calib", patch_size=patch), torch.float64)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww))).double()
    out = O.forward(x, sd, patch)
    assert np.abs(out["mask_logits"][:, -1].numpy() - g["logits_last_f64"]).max() <= 1e-10
    assert np.abs(out["objectness"].numpy() - g["objectness_f64"]).max() <= 1e-12


def test_oracle_3d_path_matches_reference_vectors():
    """forward_3d (return_intermediate=False, use_binary_classifier=False) against the real reference's output."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "forward3d_p16_224_calib.npz"))
    patch, B, Hh, Ww, wseed, xseed, nthreads = [int(v) for v in g["meta"]]
    torch.set_num_threads(min(nthreads, os.cpu_count() or 1))
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch, use_binary_classifier=False)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    out = O.forward_3d(x, sd, patch)
    scale = float(g["logit_absmax"])
    assert np.abs(out["mask_pred"].numpy() - g["mask_pred"]).max() <= 2e-6 * scale + 1e-6
    assert np.abs(out["features"].numpy() - g["features"]).max() <= 2e-5
    o64 = O.forward_3d(x.double(), O.cast_state(sd, torch.float64), patch)
    assert np.abs(o64["mask_pred"].numpy() - g["mask_pred_f64"]).max() <= 1e-10


def test_oracle_ffn_mask_head_matches_reference_vectors():
    """forward_ffn_head (return_intermediate=True, use_binary_classifier=False; maskformer.py:225) against the real
    reference's output: sigmoid(einsum(ffn(queries), up)), no objectness."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ffnhead_p16_224_soft.npz"))
    patch, B, Hh, Ww, wseed, xseed, nthreads = [int(v) for v in g["meta"]]
    torch.set_num_threads(min(nthreads, os.cpu_count() or 1))
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch, use_binary_classifier=False)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    out = O.forward_ffn_head(x, sd, patch)
    assert "objectness" not in out
    scale = float(g["logit_absmax"])
    assert np.abs(out["mask_logits"].numpy() - g["mask_logits"]).max() <= 2e-6 * scale + 1e-6
    assert np.abs(out["mask_pred"].numpy() - g["mask_pred"]).max() <= 1e-6
    assert np.abs(out["features"].numpy() - g["features"]).max() <= 2e-5
    o64 = O.forward_ffn_head(x.double(), O.cast_state(sd, torch.float64), patch)
    assert np.abs(o64["mask_logits"].numpy() - g["mask_logits_f64"]).max() <= 1e-10


def test_oracle_prenorm_decoder_matches_reference_vectors():
    """normalize_before=True (TransformerDecoderLayer.forward_pre, transformer_decoder.py:299-327) against the real
    reference's output."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "prenorm_p16_224_calib.npz"))
    patch, B, Hh, Ww, wseed, xseed, nthreads = [int(v) for v in g["meta"]]
    torch.set_num_threads(min(nthreads, os.cpu_count() or 1))
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    out = O.forward(x, sd, patch, normalize_before=True)
    scale = float(g["logit_absmax"])
    assert np.abs(out["mask_logits"].numpy() - g["mask_logits"]).max() <= 2e-6 * scale + 1e-6
    assert np.abs(out["objectness"].numpy() - g["objectness"]).max() <= 2e-6
    assert np.abs(out["features"].numpy() - g["features"]).max() <= 2e-5
    o64 = O.forward(x.double(), O.cast_state(sd, torch.float64), patch, normalize_before=True)
    assert np.abs(o64["mask_logits"].numpy() - g["mask_logits_f64"]).max() <= 1e-10



def test_oracle_scale_factor_matches_reference_vectors():
    """scale_factor 1 and 4 (maskformer.py:23,161) against the real reference's outputs (tests/golden/scalefactor_*.npz)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "scalefactor_p16_calib.npz"))
    patch, B, wseed, xseed, nthreads = [int(v) for v in g["meta"]]
    torch.set_num_threads(min(nthreads, os.cpu_count() or 1))
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch)
    for tag in [str(t) for t in g["cases"]]:
        hw, sf = tag.split("_sf")
        Hh, Ww = (int(v) for v in hw.split("x"))
        x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
        out = O.forward(x, sd, patch, scale_factor=int(sf))
        assert out["mask_logits"].shape[-2:] == (int(sf) * Hh // patch, int(sf) * Ww // patch)
        assert np.abs(out["mask_logits"][:, -1].numpy() - g[f"logits_last_{tag}"]).max() <= 2e-6 * float(g[f"logit_absmax_{tag}"]) + 1e-6
        assert np.abs(out["objectness"].numpy() - g[f"objectness_{tag}"]).max() <= 2e-6
        o64 = O.forward(x.double(), O.cast_state(sd, torch.float64), patch, scale_factor=int(sf))
        assert np.abs(o64["mask_logits"][:, -1].numpy() - g[f"logits_last_f64_{tag}"]).max() <= 1e-10
