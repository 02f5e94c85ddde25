"""Mirror of the model-factory part of the reference's ``utils/misc.py`` (get_model :163-188, set_seeds)."""
import random
from argparse import Namespace
from typing import Optional

import numpy as np
import torch

from .maskformer import MaskFormer


def get_model(arch: str, patch_size: Optional[int] = None, training_method: Optional[str] = None,
              configs: Optional[Namespace] = None, **kwargs):
    """utils/misc.py:163-188 for ``arch == "maskformer"``: reads the same config keys.  Unlike the reference it never
    fetches DINO weights from the network (utils/misc.py:196,243) - the SelfMask checkpoint overwrites them."""
    if arch == "maskformer":
        assert configs is not None
        return MaskFormer(
            n_queries=configs.n_queries,
            n_decoder_layers=configs.n_decoder_layers,
            learnable_pixel_decoder=configs.learnable_pixel_decoder,
            lateral_connection=configs.lateral_connection,
            return_intermediate=configs.loss_every_decoder_layer,
            scale_factor=configs.scale_factor,
            abs_2d_pe_init=configs.abs_2d_pe_init,
            use_binary_classifier=configs.use_binary_classifier,
            arch=configs.arch,
            training_method=configs.training_method,
            patch_size=configs.patch_size,
        )
    raise ValueError(f"{arch} is not on the MI355X hot path; only arch='maskformer' is implemented "
                     f"(reference choices: maskformer, resnet50, vit, dino)")


def set_seeds(seed: int = 0) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
