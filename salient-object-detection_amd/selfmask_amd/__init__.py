"""selfmask_amd - MI355X-native SelfMask saliency inference (ViT-S encoder -> MaskFormer decoder -> query masks).

Host side mirrors the reference's Python surface; arithmetic runs in hand-written gfx950 kernels behind the C ABI of
``include/selfmask_hip.h`` (``lib/libselfmask_hip.so``).
"""
from .maskformer import MaskFormer, load_checkpoint  # noqa: F401
from .misc import get_model, set_seeds  # noqa: F401
from .base_structure import BaseStructure  # noqa: F401
from .streams import StreamRing  # noqa: F401
from .graphs import GraphedForward  # noqa: F401
from .bilateral_solver import (bilateral_solver_output, bilateral_solver_output_device,  # noqa: F401
                               bilateral_solver_batch_device)
from .state_layout import state_shapes, synthetic_state_dict, synthetic_images  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):  # lazy: `python -m selfmask_amd.evaluator` must not find the module pre-imported
    if name == "Evaluator":
        from .evaluator import Evaluator
        return Evaluator
    if name == "MaskGenerator":
        from .mask_generator import MaskGenerator
        return MaskGenerator
    if name == "SelfMaskInference":
        from .inference import SelfMaskInference
        return SelfMaskInference
    raise AttributeError(name)
