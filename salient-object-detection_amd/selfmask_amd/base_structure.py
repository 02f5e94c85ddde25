"""Mirror of the reference's ``base_structure.BaseStructure._forward`` (base_structure.py:7-24)."""
from typing import Dict, Optional

import torch


class BaseStructure:
    def __init__(self, model: callable, visualizer: Optional[callable] = None,
                 device: torch.device = torch.device("cuda:0")):
        self.device = device
        self.model = model
        self.visualizer = visualizer

    def _forward(self, dict_data: dict, encoder_only: bool = False, skip_decoder: bool = False,
                 device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
        """base_structure.py:18-24.  ``device=`` is accepted because the compiled evaluator passes it
        (evaluator.pyc@L194) although the reference signature lacks it (a latent TypeError there)."""
        dev = device if device is not None else self.device
        x = dict_data['x'].to(dev)
        fwd = getattr(self, "_graphed", None)  # graphs.GraphedForward installed by the Evaluator (plain forwards only)
        if fwd is not None and not encoder_only and not skip_decoder:
            return fwd(x)
        return self.model(x, encoder_only=encoder_only, skip_decoder=skip_decoder)
