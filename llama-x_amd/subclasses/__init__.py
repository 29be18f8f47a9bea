"""Tensor-subclass surface of the reference (`from subclasses import quantize_linear_`), HIP-backed.

Mirrors /root/reference/subclasses/__init__.py:6-13.
"""
from torch import nn

from .int8 import Int8LinearWeight, quantize_int8_rowwise  # noqa: F401
from .int8_mm import int8_mm_dequant  # noqa: F401

_QUANTIZERS = {"int8": Int8LinearWeight.from_float}


def quantize_linear_(model: nn.Module, quantize: str | None, **kwargs):
    """Swap the weight of every ``nn.Linear`` under ``model`` for a frozen quantised Parameter (in place)."""
    if quantize is None:
        return
    make = _QUANTIZERS[quantize]  # KeyError for unknown schemes, as in the reference
    for mod in model.modules():
        if isinstance(mod, nn.Linear):
            mod.weight = nn.Parameter(make(mod.weight.detach(), **kwargs), requires_grad=False)
