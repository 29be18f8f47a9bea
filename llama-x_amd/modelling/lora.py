"""LoRA / DoRA adapters by in-place class swap (API of /root/reference/modelling/lora.py:8-62).

``LoRALinear.forward`` runs F.linear(x, W, b) + x @ A^T @ B^T * (alpha/r) as ONE fused GEMM launch: x @ A^T comes
from the skinny MFMA kernel and rides into the base GEMM as a 64-wide K-extension against s*B
(llama-x_amd/csrc/{skinny,gemm_bf16}.hip).  Inside a transformer layer the fused block functions pick the adapter
up from the module (llx/ops.py:LinearPlan), so this forward only runs for stand-alone calls.
"""
import torch
from torch import Tensor, nn

from llx import ops


def apply_linear_adapter_(model: nn.Module, adapter: str | None, **kwargs):
    if adapter is None:
        return
    target = {"lora": LoRALinear, "dora": DoRALinear}[adapter]
    for mod in model.modules():
        if isinstance(mod, nn.Linear):
            mod.__class__ = target
            mod.init_adapter(**kwargs)


class LoRALinear(nn.Linear):
    def init_adapter(self, rank: int = 8, alpha: float = 8.0) -> None:
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)
        self.rank, self.alpha = rank, alpha
        self.scale = self.alpha / self.rank
        if rank > 0:
            # for an Int8LinearWeight `.dtype` is the dtype of its scale (subclasses/int8.py)
            kw = dict(dtype=self.weight.dtype, device=self.weight.device)
            self.lora_a = nn.Parameter(torch.empty(rank, self.in_features, **kw))
            self.lora_b = nn.Parameter(torch.empty(self.out_features, rank, **kw))
            nn.init.kaiming_normal_(self.lora_a, a=5**0.5)
            nn.init.zeros_(self.lora_b)

    def extra_repr(self):
        return f"{super().extra_repr()}, rank={self.rank}, alpha={self.alpha}"

    def forward(self, x: Tensor):
        return ops.linear(x, self)


class DoRALinear(LoRALinear):
    """Weight-decomposed LoRA: (W x + s B A x) * m / ||W + s B A||_row + bias (reference modelling/lora.py:47-62).

    The row norm is evaluated on the device without the reference's dense [out, in] temporary (csrc/dora.hip: ||W_n||^2 cached,
    cross term through the skinny MFMA kernel); the rescale, the bias and the gradient of ``m`` are HIP kernels behind
    ``llx.ops.LinearPlan`` - stand-alone and inside the fused transformer blocks alike."""

    def init_adapter(self, rank: int = 8, alpha: float = 8.0) -> None:
        super().init_adapter(rank, alpha)
        if self.rank > 0:
            self.m = nn.Parameter(self.weight.norm(p=2, dim=1))

    def forward(self, x: Tensor):
        return ops.linear(x, self)
