"""Drop-in for the reference's ``modelling`` package (same exports as /root/reference/modelling/__init__.py:1-3)."""
from .audio import AudioConfig, LlamaAudio
from .llama import Llama, LlamaConfig
from .lora import DoRALinear, LoRALinear, apply_linear_adapter_

__all__ = ["AudioConfig", "LlamaAudio", "Llama", "LlamaConfig", "DoRALinear", "LoRALinear", "apply_linear_adapter_"]
