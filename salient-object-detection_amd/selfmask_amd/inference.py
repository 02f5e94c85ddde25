"""Mirror of the reference's serving class ``SelfMaskInference`` (app.py:161-347), the call pattern behind POST /predict:
``T.Resize((224, 224)) -> ToTensor -> Normalize -> BaseStructure._forward -> last decoder layer -> arg-max objectness ->
clip(mask, 0, 1)`` at batch 1.

On the MI355X the whole of it after the image decode is device work: the Pillow-exact resize + normalisation kernels
(pipeline.py), the forward replayed from ONE captured hipGraph (batch 1 is launch-bound: ~170 kernels of a few
microseconds), and a selection kernel - one small D2H copy of (index, 20 scores, mask) leaves the GPU.  The web shell
around it (Flask, base64 PNG encoding, LANCZOS resize to the upload's size, jet heat map) is product code outside the hot
path: ``predict()`` still returns the reference's response keys, built on the host from ``predict_tensors()``.
"""
import base64
import threading
from argparse import Namespace
from io import BytesIO
from typing import Optional, Union

import numpy as np
import torch
from PIL import Image

from . import _native as N
from .base_structure import BaseStructure
from .graphs import GraphedForward
from .maskformer import load_checkpoint
from .misc import get_model
from .pipeline import preprocess_on_device


class SelfMaskInference:
    def __init__(self, model_path: Optional[str], config_path: Union[str, dict, Namespace], device: Optional[torch.device] = None,
                 model: Optional[torch.nn.Module] = None, hip_graph: bool = True):
        """app.py:162-211.  ``config_path``: the reference's YAML (or an already parsed dict / Namespace); ``model_path``: a
        checkpoint in either of the reference's forms ({'model': state_dict} as app.py:185-186 expects, or a raw
        state_dict).  ``model=`` injects a ready module instead (tests, benchmarks: there is no checkpoint offline)."""
        if isinstance(config_path, str):
            import yaml
            with open(config_path, "r") as f:
                config_path = yaml.safe_load(f)
        self.config = Namespace(**config_path) if isinstance(config_path, dict) else config_path
        self.device = device if device is not None else torch.device("cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError("SelfMaskInference (MI355X) needs a HIP device; there is no CPU fallback")
        if model is None:
            model = get_model(arch="maskformer", configs=self.config)
            if model_path is not None:
                load_checkpoint(model, model_path)
        self.model = model.to(self.device).eval()
        self.base_structure = BaseStructure(model=self.model, device=self.device)
        self.input_size = 224  # T.Resize((224, 224)), app.py:199
        # batch 1, one shape: capture at the first call, replay ever after
        self.base_structure._graphed = GraphedForward(self.model, enabled=hip_graph, max_graphs=2, admit_after=0)
        # The replayed graph owns ONE static input and ONE set of outputs per (shape, stream); Flask's server is threaded
        # (app.py:3927), so requests are serialised from the copy into the static input to the last D2H copy
        self._lock = threading.Lock()
        self._out = None  # (shape key, device result buffer, pinned host copy)

    # ---- host: whatever arrives -> (H, W, 3) uint8 ------------------------------------------------------------------------
    @staticmethod
    def _to_rgb_array(image) -> np.ndarray:
        """app.py:215-219: a werkzeug FileStorage (anything with ``.stream``), a file object / path, a PIL image or an
        array - converted to RGB."""
        if isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3:
            # already what Image.fromarray(image).convert("RGB") would hand back, byte for byte: the round trip through Pillow
            # (encode + decode of the raw bytes, 0.13 ms for 300 x 400) was paid twice per request
            return np.ascontiguousarray(image)
        if hasattr(image, "stream"):
            image = Image.open(image.stream)
        elif isinstance(image, (str, bytes)) or hasattr(image, "read"):
            image = Image.open(image)
        elif isinstance(image, np.ndarray):
            image = Image.fromarray(image)
        return np.asarray(image.convert("RGB"), np.uint8)

    def preprocess_image(self, image) -> torch.Tensor:
        """app.py:213-238 -> (1, 3, 224, 224) on the device; resize / ToTensor / Normalize run in HIP kernels."""
        return preprocess_on_device([self._to_rgb_array(image)], self.input_size, self.device, pinned=True)

    # ---- device: the hot path -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict_tensors(self, image) -> dict:
        """The arithmetic of ``predict`` (app.py:241-284): {"best_idx", "objectness_scores" (nq,), "mask" (2g, 2g) in [0, 1]}."""
        rgb = self._to_rgb_array(image)  # decode outside the lock: host work of the request itself
        with self._lock:
            return self._predict_locked(rgb)

    def _predict_locked(self, rgb: np.ndarray) -> dict:
        x = self.preprocess_image(rgb)
        out = self.base_structure._forward({"x": x})
        mask_pred, obj = out["mask_pred"], out.get("objectness")
        if obj is None:
            raise RuntimeError("the serving path selects by objectness (app.py:268-276): use_binary_classifier=True")
        last, last_obj = mask_pred[:, -1], obj[:, -1, :, 0]
        nq, h, w = last.shape[1:]
        # one device buffer [best index | nq scores | h*w mask] and ONE copy into page-locked memory: the request ends with a
        # single stream synchronisation instead of three blocking device->host copies
        key = (nq, h, w)
        if self._out is None or self._out[0] != key:
            dev_buf = torch.empty(1 + nq + h * w, dtype=torch.float32, device=self.device)
            self._out = (key, dev_buf, torch.empty(1 + nq + h * w, dtype=torch.float32).pin_memory())
        _, dev_buf, host_buf = self._out
        N.check(N.load().sm_pick_mask_f32(last.data_ptr(), last.stride(0), last_obj.data_ptr(), last_obj.stride(0),
                                          dev_buf[1 + nq:].data_ptr(), dev_buf[:1].data_ptr(), 1, nq, h * w,
                                          torch.cuda.current_stream(self.device).cuda_stream), "sm_pick_mask_f32")
        dev_buf[1:1 + nq].copy_(last_obj[0])
        host_buf.copy_(dev_buf, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()  # the request is done
        out_h = host_buf.numpy()
        return {"best_idx": int(out_h[:1].view(np.int32)[0]), "objectness_scores": out_h[1:1 + nq].copy(),
                "mask": out_h[1 + nq:].reshape(h, w).copy()}

    # ---- host: the reference's response ---------------------------------------------------------------------------------------
    def predict(self, image) -> dict:
        """app.py:241-347: same keys ('original', 'mask', 'heatmap' as base64 PNG data URLs, 'objectness_scores')."""
        rgb = self._to_rgb_array(image)
        if hasattr(image, "stream"):
            image.stream.seek(0)
        t = self.predict_tensors(rgb)
        original = Image.fromarray(rgb)
        mask_img = Image.fromarray((t["mask"] * 255).astype(np.uint8)).resize(original.size, Image.Resampling.LANCZOS)
        heat = None
        try:  # the jet colour map comes from matplotlib in the reference (app.py:296-304); optional here
            import matplotlib.pyplot as plt
            from PIL import ImageEnhance
            rgba = (plt.get_cmap("jet")(np.array(mask_img) / 255.0) * 255).astype(np.uint8)
            heat_img = Image.fromarray(rgba).convert("RGBA").resize(original.size, Image.Resampling.LANCZOS)
            heat = ImageEnhance.Brightness(Image.blend(original.convert("RGBA"), heat_img, alpha=0.5)).enhance(1.1)
        except ImportError:
            pass

        def url(img):
            if img is None:
                return None
            buf = BytesIO()
            img.save(buf, format="PNG")
            return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode()

        return {"original": url(original), "mask": url(mask_img), "heatmap": url(heat),
                "objectness_scores": t["objectness_scores"], "best_idx": t["best_idx"]}
