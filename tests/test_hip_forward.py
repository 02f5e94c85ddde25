"""Whole-forward parity of the HIP path (through the C ABI) against (a) the vectors the REAL reference produced
(tests/golden, made by oracle/gen_golden.py) and (b) the CPU oracle run on this box in fp32 and fp64."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import selfmask_oracle as O  # noqa: E402  (checker only)
from selfmask_amd import MaskFormer, synthetic_state_dict, synthetic_images  # noqa: E402
import _ledger as ledger  # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "forward_*.npz")))

# BASELINE.json north_star: "within 1e-4 max-abs on logits".  The gate, as asserted here:
#   * fixtures whose fp32 reference is itself within 5e-5 of its own fp64 evaluation (the "calib" checkpoints,
#     |logit| <= 16): |hip - ref32| <= 1e-4, flat;
#   * the others (unit decoder.norm gain: |logit| up to 66, where the REFERENCE's fp32 result is 0.8-1.6e-4 away from
#     its fp64 evaluation, f32_vs_f64_maxabs in the fixtures): |hip - ref64| <= |ref32 - ref64| - the HIP path must be
#     at least as close to the fp64 truth as the reference's own fp32 arithmetic is, no slack - and therefore
#     |hip - ref32| <= 2 |ref32 - ref64| by the triangle inequality.
# Every case records |logit|max, hip-ref32, hip-ref64, ref32-ref64, the bound used, thresholded-pixel flips and
# arg-max agreement in gpurun_out/parity_ledger.json (committed as profiles/r02_parity.json).
ABS_TOL = 1e-4
STRICT_BELOW = 5e-5


def _tol(scale):
    return ABS_TOL * max(1.0, scale / 16.0)


MODES = ["w16", "f16x2", "fp32"]  # single-accumulator W16 weights (default) / two-accumulator split / exact fp32 MFMA


def _model(patch, wseed, style, mode="w16"):
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True,
                   use_binary_classifier=True, gemm_mode=mode)
    m.load_state_dict(synthetic_state_dict(wseed, style, patch_size=patch), strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("fp", CASES, ids=[os.path.basename(c)[8:-4] for c in CASES])
def test_forward_matches_reference_vectors(fp, mode):
    g = np.load(fp)
    patch, B, Hh, Ww, wseed, xseed, _ = [int(v) for v in g["meta"]]
    m = _model(patch, wseed, str(g["style"]), mode)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww))).to(DEV)
    out = m(x, return_logits=True)
    scale = float(g["logit_absmax"])
    tol = _tol(scale)
    logits = out["mask_logits"][:, -1].cpu().numpy()
    d32 = np.abs(logits - g["logits_last"]).max()
    d64 = np.abs(logits - g["logits_last_f64"]).max()
    ref64 = float(g["f32_vs_f64_maxabs"])
    strict = ref64 <= STRICT_BELOW
    print(f"\n[{mode}] {os.path.basename(fp)}: |logit|max={scale:.1f} hip-ref32={d32:.2e} hip-ref64={d64:.2e} ref32-ref64={ref64:.2e} "
          f"rule={'strict 1e-4' if strict else 'hip-ref64 <= ref32-ref64'}")
    obj_ref = g["objectness"][:, -1, :, 0]
    obj = out["objectness"][:, -1, :, 0].cpu().numpy()
    mp = out["mask_pred"][:, -1].cpu().numpy()
    ref_bin = (1 / (1 + np.exp(-g["logits_last"].astype(np.float64)))) > 0.5
    flips = float(((mp > 0.5) != ref_bin).mean())
    ledger.record("forward_fixtures", f"{os.path.basename(fp)[8:-4]}|{mode}", {
        "logit_absmax": scale, "hip_minus_ref32": float(d32), "hip_minus_ref64": float(d64), "ref32_minus_ref64": ref64,
        "rule": "hip-ref32 <= 1e-4" if strict else "hip-ref64 <= ref32-ref64", "bound": ABS_TOL if strict else ref64,
        "pixel_flips": flips, "argmax_objectness_equal": bool((obj.argmax(1) == obj_ref.argmax(1)).all()),
        "objectness_maxabs": float(np.abs(out["objectness"].cpu().numpy() - g["objectness"]).max()),
        "queries_maxabs": float(np.abs(out["queries"].cpu().numpy() - g["queries"]).max())})
    if strict:
        assert d32 <= ABS_TOL
        assert d64 <= max(ref64, 0.5 * ABS_TOL)
    else:
        assert d64 <= ref64      # at least as close to the fp64 truth as the reference's fp32 result is
        assert d32 <= 2.0 * ref64
    assert out["mask_pred"].shape == (B, 6, 20, 2 * int(g["grid"][0]), 2 * int(g["grid"][1]))
    assert np.abs(out["objectness"].cpu().numpy() - g["objectness"]).max() <= 2e-5
    assert np.abs(out["features"].cpu().numpy() - g["features"]).max() <= 5e-5
    assert np.abs(out["queries"].cpu().numpy() - g["queries"]).max() <= 5e-5
    # selection parity: arg-max objectness query and the thresholded masks vs the reference's
    assert (obj.argmax(1) == obj_ref.argmax(1)).all()
    assert flips <= 2e-5, flips
    if "logits_all" in g:
        assert np.abs(out["mask_logits"].cpu().numpy() - g["logits_all"]).max() <= tol
        assert np.abs(out["patch_tokens"][0].cpu().numpy() - g["patch_tokens_b0"]).max() <= 5e-5


@pytest.mark.parametrize("fp", CASES, ids=[os.path.basename(c)[8:-4] for c in CASES])
def test_throughput_mode_diagnostic_ledger(fp):
    """gemm_mode="f16" (ONE f16 MFMA per product, fp32 statistics) is a diagnostic, not the product: this
    records how far it lands from the reference's vectors on every fixture (max-abs logit error, pixel flips, arg-max
    objectness, IoU of the thresholded last-layer masks to 3 d.p.) and asserts only that it is sane (finite, an f16-sized
    error - well outside the 1e-4 gate - and mostly the same masks)."""
    g = np.load(fp)
    patch, B, Hh, Ww, wseed, xseed, _ = [int(v) for v in g["meta"]]
    m = _model(patch, wseed, str(g["style"]), "f16")
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww))).to(DEV)
    out = m(x, return_logits=True)
    logits = out["mask_logits"][:, -1].cpu().numpy()
    assert np.isfinite(logits).all()
    scale = float(g["logit_absmax"])
    d32 = float(np.abs(logits - g["logits_last"]).max())
    ref_bin = g["logits_last"] >= 0
    got_bin = logits >= 0
    flips = float((got_bin != ref_bin).mean())
    inter = (got_bin & ref_bin).sum(axis=(-1, -2)).astype(np.float64)
    union = (got_bin | ref_bin).sum(axis=(-1, -2)).astype(np.float64)
    iou = float(np.where(union > 0, inter / np.maximum(union, 1), 1.0).mean())  # f16 masks vs the reference's masks
    obj_ref, obj = g["objectness"][:, -1, :, 0], out["objectness"][:, -1, :, 0].cpu().numpy()
    ledger.record("throughput_mode_f16", os.path.basename(fp)[8:-4], {
        "logit_absmax": scale, "hip_minus_ref32": d32, "pixel_flips": flips, "mask_iou_vs_reference": round(iou, 4),
        "mask_iou_is_1_to_3dp": round(iou, 3) == 1.0, "argmax_objectness_equal": bool((obj.argmax(1) == obj_ref.argmax(1)).all()),
        "objectness_maxabs": float(np.abs(obj - obj_ref).max()), "rule": "diagnostic: recorded, not gated"})
    print(f"\n[f16 throughput mode] {os.path.basename(fp)}: |logit|max={scale:.1f} hip-ref32={d32:.2e} flips={flips:.2e} iou={iou:.4f}")
    assert 1e-4 < d32 <= 0.05 * max(scale, 16.0), d32   # an f16-sized error: not the fp32-grade path, not garbage
    assert flips <= 2e-2 and iou >= 0.95


@pytest.mark.parametrize("mode", MODES)
def test_forward_vs_oracle_batch8_and_fp64_truth(mode):
    """Larger seeded batch through the oracle on this box's CPU (fp32 + fp64), strict 1e-4 on a calib checkpoint."""
    patch, B = 16, 8
    sd = synthetic_state_dict(11, "calib", patch_size=patch)
    x = torch.from_numpy(synthetic_images(4321, (B, 3, 224, 224)))
    m = _model(patch, 11, "calib", mode)
    out = m(x.to(DEV), return_logits=True)
    o32 = O.forward(x, sd, patch)
    o64 = O.forward(x.double(), O.cast_state(sd, torch.float64), patch)
    lg = out["mask_logits"].cpu()
    d32 = (lg - o32["mask_logits"]).abs().max().item()
    d64 = (lg.double() - o64["mask_logits"]).abs().max().item()
    r64 = (o32["mask_logits"].double() - o64["mask_logits"]).abs().max().item()
    print(f"\n[{mode}] B=8 calib: |logit|max={o32['mask_logits'].abs().max():.1f} hip-oracle32={d32:.2e} hip-truth64={d64:.2e} "
          f"oracle32-truth64={r64:.2e}")
    ledger.record("forward_oracle_b8_calib", mode, {"logit_absmax": float(o32["mask_logits"].abs().max()), "hip_minus_ref32": d32,
                                                    "hip_minus_ref64": d64, "ref32_minus_ref64": r64, "rule": "hip-ref32 <= 1e-4"})
    assert d32 <= ABS_TOL
    assert d64 <= max(r64, 0.5 * ABS_TOL)
    assert (out["mask_pred"].cpu() - o32["mask_pred"]).abs().max().item() <= 0.25 * ABS_TOL + 1e-6
    assert (out["objectness"].cpu() - o32["objectness"]).abs().max().item() <= 2e-5
    assert (out["features"].cpu() - o32["features"]).abs().max().item() <= 5e-5


def test_forward_is_deterministic_and_batch_invariant():
    m = _model(16, 0, "soft")
    x = torch.from_numpy(synthetic_images(99, (4, 3, 224, 224))).to(DEV)
    # with the encoder path pinned (as the Evaluator pins it): the automatic choice depends on the batch size - fused QKV +
    # attention from 16 images up, fc2 split along K below 513 token rows - and those agree to rounding, not to the bit
    m.attention_path = "unfused"
    a = m(x, return_logits=True)
    b = m(x, return_logits=True)
    assert torch.equal(a["mask_logits"], b["mask_logits"]) and torch.equal(a["objectness"], b["objectness"])
    # image i alone gives the same bits as image i inside the batch (no cross-image reduction anywhere)
    c = m(x[2:3], return_logits=True)
    assert torch.equal(c["mask_logits"][0], a["mask_logits"][2])
    # the automatic path at batch 1 (the serving path: fc2 as four K-slices summed by the LayerNorm launch) against the pinned one
    m.attention_path = "auto"
    d = m(x[2:3], return_logits=True)
    again = m(x[2:3], return_logits=True)
    assert torch.equal(d["mask_logits"], again["mask_logits"])
    err = (d["mask_logits"] - c["mask_logits"]).abs().max().item()
    assert err < 1e-4, err  # two fp32-grade orders of the same sums (measured 3e-5 on logits up to ~30)
    assert d["mask_pred"].shape == c["mask_pred"].shape and bool(d["mask_pred"].isfinite().all())


@pytest.mark.parametrize("mode", MODES)
def test_full_bench_batch_64_properties(mode):
    """BASELINE.json configs[1] at its full size (B=64, ViT-S/16, 224^2): too big for the CPU oracle in a test, so
    (1) batch invariance - images 0, 29 and 63 alone give the same bits as inside the batch of 64 although the GEMMs
    then run on other tile shapes (with the encoder attention path pinned: "auto" takes the fused kernel at B = 64 and the
    GEMM + attention pair at B = 1, which agree to rounding, not to the bit - checked below); (2) the oracle is run on those three images only, strict 1e-4 gate (calib weights);
    (3) two identical images in one batch give identical outputs; (4) a second call reproduces the first bit for bit."""
    patch, B = 16, 64
    sd = synthetic_state_dict(12, "calib", patch_size=patch)
    xs = synthetic_images(777, (B, 3, 224, 224))
    xs[40] = xs[7]
    x = torch.from_numpy(xs)
    m = _model(patch, 12, "calib", mode)
    m.attention_path = "fused"
    out = m(x.to(DEV), return_logits=True)
    again = m(x.to(DEV), return_logits=True)
    assert torch.equal(out["mask_logits"], again["mask_logits"]) and torch.equal(out["objectness"], again["objectness"])
    assert torch.equal(out["mask_logits"][40], out["mask_logits"][7]) and torch.equal(out["features"][40], out["features"][7])
    pick = [0, 29, 63]
    for i in pick:
        one = m(x[i:i + 1].to(DEV), return_logits=True)
        assert torch.equal(one["mask_logits"][0], out["mask_logits"][i]), i
        assert torch.equal(one["objectness"][0], out["objectness"][i]), i
    # B = 1 on the automatic path: two-launch attention, folded pre-norms, fc2 as four K-slices - the same result to rounding
    m.attention_path = "auto"
    one = m(x[29:30].to(DEV), return_logits=True)
    assert (one["mask_logits"][0] - out["mask_logits"][29]).abs().max().item() <= 5e-5
    m.attention_path = "unfused"
    pair = m(x[29:30].to(DEV), return_logits=True)
    assert (pair["mask_logits"] - one["mask_logits"]).abs().max().item() <= 5e-5
    o32 = O.forward(x[pick], sd, patch)
    d = (out["mask_logits"][pick].cpu() - o32["mask_logits"]).abs().max().item()
    print(f"\n[{mode}] B=64 calib, images {pick}: hip-oracle32={d:.2e}")
    ledger.record("forward_b64_calib_images_0_29_63", mode, {"hip_minus_ref32": d, "rule": "hip-ref32 <= 1e-4"})
    assert d <= ABS_TOL
    assert (out["objectness"][pick].cpu() - o32["objectness"]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("patch,size,B", [(8, 224, 16), (8, 224, 32), (16, 384, 32)], ids=["vit_s8_224_b16", "vit_s8_224_b32", "vit_s16_384_b32"])
def test_other_bench_shapes_at_full_size(patch, size, B):
    """The two other shapes bench.py times, AT the batch it times them with: ViT-S/8 224^2 x 16 and x 32 (N = 785: the
    shipped checkpoint's patch size, M = 12 560 / 25 120 token rows; round 3 benched 16, round 4 benches 32) and ViT-S/16 384^2 x 32 (N = 577: BASELINE configs[2], M = 18 464) - the 256 x 256
    / 256 x 128 tiles at those M and attention_f16x2 at those N.  Same properties as the batch-64 test: a second call gives the
    same bits, twin images give twin outputs, three picked images alone give the bits they have inside the batch, and those three
    meet the strict 1e-4 gate against the CPU oracle (calib weights)."""
    sd = synthetic_state_dict(12, "calib", patch_size=patch)
    xs = synthetic_images(778, (B, 3, size, size))
    xs[B - 3] = xs[2]
    x = torch.from_numpy(xs)
    m = _model(patch, 12, "calib")
    # the large-batch kernel set, pinned as the Evaluator pins it ("fused": LayerNorm launches, no folded pre-norms; with more than 208
    # tokens the attention is the GEMM + attention_f16x2 pair in both sets) - what "auto" picks at these batch sizes
    m.attention_path = "fused"
    out = m(x.to(DEV), return_logits=True)
    again = m(x.to(DEV), return_logits=True)
    assert torch.equal(out["mask_logits"], again["mask_logits"]) and torch.equal(out["objectness"], again["objectness"])
    assert torch.equal(out["mask_logits"][B - 3], out["mask_logits"][2]) and torch.equal(out["features"][B - 3], out["features"][2])
    pick = [0, B // 2 - 1, B - 1]
    for i in pick:
        one = m(x[i:i + 1].to(DEV), return_logits=True)
        assert torch.equal(one["mask_logits"][0], out["mask_logits"][i]), i
        assert torch.equal(one["objectness"][0], out["objectness"][i]), i
    m.attention_path = "auto"  # one image on the automatic path: the small-batch set (folded pre-norms) - the same result to rounding
    auto_b = m(x.to(DEV), return_logits=True)
    assert torch.equal(auto_b["mask_logits"], out["mask_logits"])  # and at the bench batch "auto" IS the pinned set
    one = m(x[pick[1]:pick[1] + 1].to(DEV), return_logits=True)
    assert (one["mask_logits"][0] - out["mask_logits"][pick[1]]).abs().max().item() <= 5e-5
    o32 = O.forward(x[pick], sd, patch)
    d = (out["mask_logits"][pick].cpu() - o32["mask_logits"]).abs().max().item()
    print(f"\n[w16] ViT-S/{patch} {size}^2 B={B} calib, images {pick}: hip-oracle32={d:.2e}")
    ledger.record(f"forward_vit_s{patch}_{size}_b{B}_calib_images", "w16", {"hip_minus_ref32": d, "rule": "hip-ref32 <= 1e-4", "images": pick})
    assert d <= ABS_TOL
    assert (out["objectness"][pick].cpu() - o32["objectness"]).abs().max().item() <= 2e-5


def test_hip_graph_replay_gives_the_eager_bits():
    """graphs.GraphedForward: the first two calls are eager, the third captures + replays, later ones replay; a new
    batch through the captured graph equals the eager forward bit for bit, on the default stream and on a side stream."""
    from selfmask_amd import GraphedForward
    m = _model(16, 3, "soft")
    xs = [torch.from_numpy(synthetic_images(200 + i, (3, 3, 224, 224))).to(DEV) for i in range(5)]
    eager = [{k: v.clone() for k, v in m(x).items()} for x in xs]
    for stream in (None, torch.cuda.Stream()):
        g = GraphedForward(m)
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for x, e in zip(xs, eager):
                out = g(x)
                for k in e:
                    assert torch.equal(out[k], e[k]), k
        torch.cuda.synchronize()
        assert g.failed is None and g.captures == 1 and g.replays == 3
    # a shape seen once or twice stays eager
    g = GraphedForward(m)
    g(xs[0][:2])
    g(xs[0][:2])
    assert g.captures == 0 and g.replays == 0
    # bounded number of live graphs: two shapes alternating through a one-graph cache keep giving the eager bits
    g = GraphedForward(m, max_graphs=1, admit_after=1)
    for rep in range(4):
        for x, e in ((xs[0], eager[0]), (xs[1][:2], None)):
            out = g(x)
            if e is not None:
                assert torch.equal(out["mask_pred"], e["mask_pred"])
    assert g.failed is None and g.captures >= 2 and len(g.policy) == 1


def test_hip_graph_follows_new_weights():
    """A captured graph holds raw pointers to the packed (split) weight copies; load_state_dict rebuilds those.  The
    next call through GraphedForward must give the NEW weights' eager result, not replay stale pointers."""
    from selfmask_amd import GraphedForward
    m = _model(16, 3, "soft")
    x = torch.from_numpy(synthetic_images(321, (2, 3, 224, 224))).to(DEV)
    g = GraphedForward(m, admit_after=0)
    first = {k: v.clone() for k, v in g(x).items()}
    assert g.captures == 1
    m.load_state_dict(synthetic_state_dict(8, "calib", patch_size=16), strict=True)
    eager_new = {k: v.clone() for k, v in m(x).items()}
    assert not torch.equal(eager_new["mask_pred"], first["mask_pred"])
    out = g(x)   # generation changed: old graph destroyed, this call re-captures under the new weights
    for k in eager_new:
        assert torch.equal(out[k], eager_new[k]), k
    assert g.captures == 2 and g.failed is None


@pytest.mark.parametrize("patch,B,H,W,nq,L,seed", [
    (16, 1, 224, 224, 100, 3, 1),   # the reference's default n_queries, fewer decoder layers, 4 query blocks in cross-attention
    (16, 5, 97, 211, 7, 1, 2),      # ragged image (zero-padded to 112 x 224), odd batch, single decoder layer
    (8, 3, 72, 88, 20, 6, 3),       # ViT-S/8, 99 tokens (not a multiple of 4: literal up-sample + einsum order)
    (16, 2, 32, 32, 20, 2, 4),      # four tokens per image: every GEMM / attention tile is mostly padding
    (8, 1, 250, 130, 33, 4, 5),     # off-grid position embedding (bicubic), 33 queries = two query blocks
])
def test_forward_other_shapes_and_model_sizes_vs_oracle(patch, B, H, W, nq, L, seed):
    """Edge shapes the bench never sees, calib weights, strict 1e-4 against the CPU oracle (fp32) on this box."""
    sd = synthetic_state_dict(seed, "calib", n_queries=nq, patch_size=patch, n_decoder_layers=L)
    m = MaskFormer(n_queries=nq, patch_size=patch, n_decoder_layers=L, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    x = torch.from_numpy(synthetic_images(900 + seed, (B, 3, H, W)))
    out = m(x.to(DEV), return_logits=True)
    ref = O.forward(x, sd, patch, n_layers=L)
    scale = ref["mask_logits"].abs().max().item()
    d = (out["mask_logits"].cpu() - ref["mask_logits"]).abs().max().item()
    print(f"\n P{patch} B={B} {H}x{W} nq={nq} L={L}: |logit|max={scale:.1f} hip-oracle32={d:.2e}")
    ledger.record("forward_edge_shapes_calib", f"P{patch}_B{B}_{H}x{W}_nq{nq}_L{L}", {"logit_absmax": scale, "hip_minus_ref32": d,
                                                                                   "rule": "hip-ref32 <= 1e-4"})
    assert out["mask_pred"].shape == ref["mask_pred"].shape
    assert scale <= 16.0 and d <= ABS_TOL
    assert (out["objectness"].cpu() - ref["objectness"]).abs().max().item() <= 2e-5
    assert (out["features"].cpu() - ref["features"]).abs().max().item() <= 5e-5


def test_encoder_only_and_3d_path():
    m = _model(16, 0, "soft")
    x = torch.from_numpy(synthetic_images(5, (2, 3, 224, 224))).to(DEV)
    full = m(x, return_logits=True)
    enc = m(x, encoder_only=True)
    assert enc["patch_tokens"].shape == (2, 14, 14, 384)
    assert torch.equal(enc["patch_tokens"].reshape(2, 196, 384), full["patch_tokens"])
    m3 = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=False,
                    use_binary_classifier=False)
    sd = synthetic_state_dict(0, "soft", patch_size=16, use_binary_classifier=False)
    m3.load_state_dict(sd, strict=True)
    o3 = m3.to(DEV)(x)
    assert set(o3.keys()) == {"mask_pred", "features"} and o3["mask_pred"].shape == (2, 20, 28, 28)


@pytest.mark.parametrize("mode", MODES)
def test_3d_path_values_match_reference_vectors(mode):
    """return_intermediate=False, use_binary_classifier=False (maskformer.py:219-220,246-249): un-sigmoided last-layer
    logits + features against the REAL reference's output (tests/golden/forward3d_*.npz, oracle/gen_golden.py) and
    against the oracle's restatement of that path - values, not just shapes."""
    g = np.load(os.path.join(GOLD, "forward3d_p16_224_calib.npz"))
    patch, B, Hh, Ww, wseed, xseed, _ = [int(v) for v in g["meta"]]
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch, use_binary_classifier=False)
    m3 = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=False,
                    use_binary_classifier=False, gemm_mode=mode)
    m3.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    o3 = m3.to(DEV)(x.to(DEV))
    assert set(o3.keys()) == {"mask_pred", "features"}
    got = o3["mask_pred"].cpu().numpy()
    d32, d64 = np.abs(got - g["mask_pred"]).max(), np.abs(got - g["mask_pred_f64"]).max()
    ref64 = float(g["f32_vs_f64_maxabs"])
    ora = O.forward_3d(x, sd, patch)
    dora = (o3["mask_pred"].cpu() - ora["mask_pred"]).abs().max().item()
    print(f"\n[{mode}] 3-D path: |logit|max={float(g['logit_absmax']):.1f} hip-ref32={d32:.2e} hip-ref64={d64:.2e} ref32-ref64={ref64:.2e} hip-oracle={dora:.2e}")
    ledger.record("forward_3d_path", mode, {"logit_absmax": float(g["logit_absmax"]), "hip_minus_ref32": float(d32),
                                            "hip_minus_ref64": float(d64), "ref32_minus_ref64": ref64, "rule": "hip-ref32 <= 1e-4"})
    assert ref64 <= STRICT_BELOW and d32 <= ABS_TOL and d64 <= max(ref64, 0.5 * ABS_TOL) and dora <= ABS_TOL
    assert np.abs(o3["features"].cpu().numpy() - g["features"]).max() <= 5e-5


@pytest.mark.parametrize("mode", MODES)
def test_scale_factor_1_and_4_match_reference_vectors(mode):
    """scale_factor (maskformer.py:23,161; YAML key): F.interpolate(scale_factor=s) in the pixel decoder - s = 1 and 4 against the
    REAL reference's outputs (tests/golden/scalefactor_p16_calib.npz): a grid whose einsum commutes with the up-sampling
    (14 x 12 tokens) and one that takes the literal order (14 x 13)."""
    g = np.load(os.path.join(GOLD, "scalefactor_p16_calib.npz"))
    patch, B, wseed, xseed, _ = [int(v) for v in g["meta"]]
    for tag in [str(t) for t in g["cases"]]:
        hw, sf = tag.split("_sf")
        Hh, Ww = (int(v) for v in hw.split("x"))
        m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True,
                       scale_factor=int(sf), gemm_mode=mode)
        m.load_state_dict(synthetic_state_dict(wseed, str(g["style"]), patch_size=patch), strict=True)
        x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww))).to(DEV)
        out = m.to(DEV)(x, return_logits=True)
        assert out["mask_pred"].shape == (B, 6, 20, int(sf) * Hh // patch, int(sf) * Ww // patch)
        d32 = np.abs(out["mask_logits"][:, -1].cpu().numpy() - g[f"logits_last_{tag}"]).max()
        d64 = np.abs(out["mask_logits"][:, -1].cpu().numpy() - g[f"logits_last_f64_{tag}"]).max()
        ref64 = float(g[f"f32_vs_f64_maxabs_{tag}"])
        ledger.record("forward_scale_factor", f"{tag}|{mode}", {"hip_minus_ref32": float(d32), "hip_minus_ref64": float(d64),
                                                                 "ref32_minus_ref64": ref64, "rule": "hip-ref32 <= 1e-4"})
        assert ref64 <= STRICT_BELOW and d32 <= ABS_TOL and d64 <= max(ref64, 0.5 * ABS_TOL), (tag, d32, d64)
        assert np.abs(out["objectness"].cpu().numpy() - g[f"objectness_{tag}"]).max() <= 2e-5
        assert np.abs(out["features"].cpu().numpy() - g[f"features_{tag}"]).max() <= 5e-5
    with pytest.raises(ValueError):
        MaskFormer(n_queries=20, patch_size=16, scale_factor=0)


@pytest.mark.parametrize("mode", MODES)
def test_forward_ffn_mask_head_matches_reference(mode):
    """return_intermediate=True with use_binary_classifier=False (maskformer.py:59-66,225): the mask einsum takes
    ffn(queries), the dict has no objectness - against the REAL reference's output (tests/golden/ffnhead_*.npz)."""
    g = np.load(os.path.join(GOLD, "ffnhead_p16_224_soft.npz"))
    patch, B, Hh, Ww, wseed, xseed, _ = [int(v) for v in g["meta"]]
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch, use_binary_classifier=False)
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, return_intermediate=True,
                   use_binary_classifier=False, learnable_pixel_decoder=True, gemm_mode=mode)  # the flag is a no-op there too
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    o = m.to(DEV)(x.to(DEV), return_logits=True)
    assert set(m.to(DEV)(x.to(DEV)).keys()) == {"mask_pred", "features"}
    got = o["mask_logits"].cpu().numpy()
    d32, d64 = np.abs(got - g["mask_logits"]).max(), np.abs(got - g["mask_logits_f64"]).max()
    ref64 = float(g["f32_vs_f64_maxabs"])
    print(f"\n[{mode}] ffn mask head: |logit|max={float(g['logit_absmax']):.1f} hip-ref32={d32:.2e} hip-ref64={d64:.2e} ref32-ref64={ref64:.2e}")
    ledger.record("forward_ffn_mask_head", mode, {"logit_absmax": float(g["logit_absmax"]), "hip_minus_ref32": float(d32),
                                                  "hip_minus_ref64": float(d64), "ref32_minus_ref64": ref64, "rule": "hip-ref32 <= 1e-4"})
    assert ref64 <= STRICT_BELOW and d32 <= ABS_TOL and d64 <= max(ref64, 0.5 * ABS_TOL)
    assert np.abs(o["mask_pred"].cpu().numpy() - g["mask_pred"]).max() <= 3e-5
    assert np.abs(o["features"].cpu().numpy() - g["features"]).max() <= 5e-5
    # the hip result thresholds like the reference's
    assert ((o["mask_pred"].cpu().numpy() > 0.5) != (g["mask_pred"] > 0.5)).sum() <= 2


@pytest.mark.parametrize("mode", MODES)
def test_forward_prenorm_decoder_matches_reference(mode):
    """normalize_before=True: TransformerDecoderLayer.forward_pre (transformer_decoder.py:299-327) in every decoder layer -
    against the REAL reference's output (tests/golden/prenorm_*.npz)."""
    g = np.load(os.path.join(GOLD, "prenorm_p16_224_calib.npz"))
    patch, B, Hh, Ww, wseed, xseed, _ = [int(v) for v in g["meta"]]
    sd = synthetic_state_dict(wseed, str(g["style"]), patch_size=patch)
    m = MaskFormer(n_queries=20, patch_size=patch, n_decoder_layers=6, normalize_before=True, return_intermediate=True,
                   use_binary_classifier=True, gemm_mode=mode)
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synthetic_images(xseed, (B, 3, Hh, Ww)))
    o = m.to(DEV)(x.to(DEV), return_logits=True)
    got = o["mask_logits"].cpu().numpy()
    d32, d64 = np.abs(got - g["mask_logits"]).max(), np.abs(got - g["mask_logits_f64"]).max()
    ref64 = float(g["f32_vs_f64_maxabs"])
    print(f"\n[{mode}] pre-norm decoder: |logit|max={float(g['logit_absmax']):.1f} hip-ref32={d32:.2e} hip-ref64={d64:.2e} ref32-ref64={ref64:.2e}")
    ledger.record("forward_prenorm_decoder", mode, {"logit_absmax": float(g["logit_absmax"]), "hip_minus_ref32": float(d32),
                                                    "hip_minus_ref64": float(d64), "ref32_minus_ref64": ref64, "rule": "hip-ref32 <= 1e-4"})
    assert ref64 <= STRICT_BELOW and d32 <= ABS_TOL and d64 <= max(ref64, 0.5 * ABS_TOL)
    assert np.abs(o["objectness"].cpu().numpy() - g["objectness"]).max() <= 2e-5
    assert np.abs(o["features"].cpu().numpy() - g["features"]).max() <= 5e-5


def test_lateral_connection_fails_like_the_reference():
    """lateral_connection=True constructs, and forward fails on the pixel decoder's 4-D assertion (maskformer.py:160)."""
    m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True,
                   lateral_connection=True)
    m.load_state_dict(synthetic_state_dict(0, "calib", patch_size=16), strict=True)
    with pytest.raises(AssertionError):
        m.to(DEV)(torch.zeros(1, 3, 224, 224, device=DEV))


def test_cpu_input_is_refused():
    m = _model(16, 0, "soft")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))
