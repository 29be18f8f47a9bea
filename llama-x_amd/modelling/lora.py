"""LoRA / DoRA adapters, attached by swapping the class of existing nn.Linear modules (API of the reference's
modelling/lora.py:8-62: ``apply_linear_adapter_``, ``LoRALinear``, ``DoRALinear``, parameters ``lora_a`` / ``lora_b`` / ``m``).

The adapter never runs as its own matmuls here.  ``x @ A^T`` comes from the skinny MFMA kernel and rides into the base GEMM as a
64-wide K-extension against ``scale * B`` (csrc/skinny.hip, csrc/gemm_bf16.hip): one launch per linear, or per fused q|k|v / gate|up
group inside a transformer layer, where ``llx.ops.LinearPlan`` picks the factors up from the module.  DoRA's row norm is evaluated
without the dense [out, in] temporary (csrc/dora.hip: ||W_n||^2 cached, cross term through the skinny kernel); its rescale, bias and
the gradient of ``m`` are kernels behind the same plan.  The ``forward`` below only serves stand-alone calls of the module.
"""
import torch
from torch import Tensor, nn

from llx import ops

_ADAPTERS: dict[str, type] = {}


def _adapter(name: str):
    def register(cls):
        _ADAPTERS[name] = cls
        return cls

    return register


def apply_linear_adapter_(model: nn.Module, adapter: str | None, **kwargs):
    """Turn every nn.Linear under ``model`` into the named adapter class in place (None: leave the model alone)."""
    if adapter is None:
        return
    cls = _ADAPTERS[adapter]
    for linear in (m for m in model.modules() if isinstance(m, nn.Linear)):
        linear.__class__ = cls
        linear.init_adapter(**kwargs)


def _freeze(*tensors):
    for t in tensors:
        if t is not None:
            t.requires_grad_(False)


@_adapter("lora")
class LoRALinear(nn.Linear):
    def init_adapter(self, rank: int = 8, alpha: float = 8.0) -> None:
        _freeze(self.weight, self.bias)
        self.rank = rank
        self.alpha = alpha
        self.scale = alpha / rank
        if rank <= 0:
            return
        # an Int8LinearWeight reports the dtype of its scale as .dtype (subclasses/int8.py): the factors follow it
        like = dict(device=self.weight.device, dtype=self.weight.dtype)
        down = torch.empty(rank, self.in_features, **like)   # A: kaiming-normal with a = sqrt(5), drawn before B is touched
        nn.init.kaiming_normal_(down, a=5**0.5)
        up = torch.zeros(self.out_features, rank, **like)    # B: zeros, so the adapted layer starts as the base layer
        self.lora_a = nn.Parameter(down)
        self.lora_b = nn.Parameter(up)

    def extra_repr(self):
        return f"{super().extra_repr()}, rank={self.rank}, alpha={self.alpha}"

    def forward(self, x: Tensor):
        return ops.linear(x, self)


@_adapter("dora")
class DoRALinear(LoRALinear):
    """Weight-decomposed LoRA: (W x + s B A x) * m / ||W + s B A||_row + bias, with ``m`` initialised to the row norms of W."""

    def init_adapter(self, rank: int = 8, alpha: float = 8.0) -> None:
        super().init_adapter(rank, alpha)
        if self.rank > 0:
            self.m = nn.Parameter(self.weight.norm(p=2, dim=1))
