"""Model-walk helpers (interface of reference model_utils.py:14-57)."""
import re
from typing import Dict, Optional

import numpy as np
import torch.nn as nn
from torch.nn.modules.conv import _ConvNd

LINEAR_LAYERS = (nn.Linear, _ConvNd)


class ForwardInterrupt(Exception):
    """Raised by InputCollector to stop the model forward after the first block's inputs are seen."""


def _to(data, **kw):
    import torch
    if isinstance(data, torch.Tensor):
        return data.to(**kw)
    if isinstance(data, (list, tuple)):
        return type(data)(_to(v, **kw) for v in data)
    if isinstance(data, dict):
        return {k: _to(v, **kw) for k, v in data.items()}
    return data


class InputCollector(nn.Module):
    """Wraps the first block: records (args, kwargs) of every call, then aborts the forward."""

    def __init__(self, module: nn.Module, cpu_offload: bool = False):
        super().__init__()
        self.module = module
        self.cpu_offload = cpu_offload
        self.input_args, self.input_kwargs = [], []

    def forward(self, *args, **kwargs):
        if self.cpu_offload:
            args, kwargs = _to(args, device="cpu"), _to(kwargs, device="cpu")
        self.input_args.append(args)
        self.input_kwargs.append(kwargs)
        raise ForwardInterrupt


def select_layers(model: nn.Module, layer_prefix: Optional[str] = "", layer_regex: str = ".*",
                  layer_classes=nn.Module) -> Dict[str, nn.Module]:
    """{dotted name: module} for modules of `layer_classes` whose name starts with the prefix and matches the regex."""
    return {n: m for n, m in model.named_modules()
            if isinstance(m, layer_classes) and n.startswith(layer_prefix) and re.search(layer_regex, n)}


def get_number_of_rows_and_cols(layer):
    return layer.weight.shape[0], int(np.prod(layer.weight.shape[1:]))
