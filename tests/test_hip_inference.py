"""The serving call pattern (selfmask_amd.SelfMaskInference, mirror of app.py:161-347) against the oracle chained the
same way: PIL resize 224 -> ToTensor -> Normalize -> forward -> last layer -> arg-max objectness -> clip(mask)."""
import io
from argparse import Namespace

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from oracle import selfmask_oracle as O  # noqa: E402  (checker only)
from selfmask_amd import MaskFormer, SelfMaskInference, synthetic_state_dict, datasets as DS  # noqa: E402

DEV = torch.device("cuda:0")
CFG = dict(n_queries=20, n_decoder_layers=6, learnable_pixel_decoder=False, lateral_connection=False,
           loss_every_decoder_layer=True, scale_factor=2, abs_2d_pe_init=False, use_binary_classifier=True,
           arch="vit_small", training_method="dino", patch_size=16)


def _oracle(rgb, sd, patch):
    im = Image.fromarray(rgb).resize((224, 224), Image.BILINEAR)
    x = np.asarray(im, np.float32) / np.float32(255.0)
    x = (x - np.asarray(DS.MEAN, np.float32)) / np.asarray(DS.STD, np.float32)
    out = O.forward(torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None], sd, patch)
    obj = out["objectness"][0, -1, :, 0]
    best = int(torch.argmax(obj))
    return best, obj.numpy(), np.clip(out["mask_pred"][0, -1, best].numpy(), 0, 1)


def test_predict_matches_oracle_and_replays_a_graph(tmp_path):
    sd = synthetic_state_dict(9, "calib", patch_size=16)
    torch.save({"model": sd, "n_epochs": 12}, tmp_path / "latest_model.pt")   # the wrapped form app.py:185-186 loads
    inf = SelfMaskInference(str(tmp_path / "latest_model.pt"), dict(CFG), device=DEV)
    rng = np.random.Generator(np.random.PCG64(3))
    for k, (h, w) in enumerate([(300, 400), (224, 224), (411, 275), (300, 400)]):
        yy, xx = np.mgrid[:h, :w]
        rgb = np.stack([(120 + 80 * np.sin(xx / 23.0 + k)), (90 + 60 * np.cos(yy / 31.0)), (60 + 0.3 * xx)], -1)
        rgb = np.clip(rgb + rng.standard_normal(rgb.shape) * 10, 0, 255).astype(np.uint8)
        got = inf.predict_tensors(rgb)
        best, obj, mask = _oracle(rgb, sd, 16)
        assert got["best_idx"] == best
        assert np.abs(got["objectness_scores"] - obj).max() <= 2e-5
        assert got["mask"].shape == (28, 28) and np.abs(got["mask"] - mask).max() <= 2.5e-5 + 1e-6
    g = inf.base_structure._graphed
    assert g.failed is None and g.captures == 1 and g.replays == 4   # batch 1, one shape: one capture, then replays


def test_predict_response_keys_and_input_kinds(tmp_path):
    m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(2, "soft", patch_size=16), strict=True)
    inf = SelfMaskInference(None, Namespace(**CFG), device=DEV, model=m)
    rgb = np.random.Generator(np.random.PCG64(1)).integers(0, 256, size=(120, 90, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="PNG")

    class Upload:  # stands for werkzeug's FileStorage: predict() reads .stream twice (app.py:216, 295-297)
        stream = io.BytesIO(buf.getvalue())

    a = inf.predict(Upload())
    b = inf.predict(Image.fromarray(rgb))
    assert set(a) >= {"original", "mask", "heatmap", "objectness_scores"} and a["mask"].startswith("data:image/png;base64,")
    assert a["mask"] == b["mask"] and a["best_idx"] == b["best_idx"] and len(a["objectness_scores"]) == 20


def test_concurrent_requests_get_their_own_results():
    """Flask serves requests on threads (app.py:3927): the shared hipGraph's static input / outputs must not interleave.
    Eight threads x six requests over four different images; every response equals the single-threaded one."""
    import threading
    m = MaskFormer(n_queries=20, patch_size=16, n_decoder_layers=6, return_intermediate=True, use_binary_classifier=True)
    m.load_state_dict(synthetic_state_dict(5, "soft", patch_size=16), strict=True)
    inf = SelfMaskInference(None, Namespace(**CFG), device=DEV, model=m)
    rng = np.random.Generator(np.random.PCG64(11))
    imgs = [rng.integers(0, 256, size=(200 + 37 * k, 260 - 21 * k, 3), dtype=np.uint8) for k in range(4)]
    want = [inf.predict_tensors(im) for im in imgs]
    errors = []

    def worker(t):
        try:
            for k in range(6):
                i = (t + k) % 4
                got = inf.predict_tensors(imgs[i])
                if got["best_idx"] != want[i]["best_idx"] or not np.array_equal(got["mask"], want[i]["mask"]) \
                        or not np.array_equal(got["objectness_scores"], want[i]["objectness_scores"]):
                    errors.append((t, k, i))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
