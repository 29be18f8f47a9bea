"""Drop-in for the reference's ``modelling`` package: the same seven public names, so host code written against the reference
(``from modelling import Llama, LlamaConfig, apply_linear_adapter_`` ...) imports unchanged.  The modules behind them run every device
operation through the HIP library (llx/), never through torch fallbacks."""
from . import audio as _audio
from . import llama as _llama
from . import lora as _lora

Llama, LlamaConfig = _llama.Llama, _llama.LlamaConfig
LlamaAudio, AudioConfig = _audio.LlamaAudio, _audio.AudioConfig
LoRALinear, DoRALinear, apply_linear_adapter_ = _lora.LoRALinear, _lora.DoRALinear, _lora.apply_linear_adapter_

__all__ = ["AudioConfig", "LlamaAudio", "Llama", "LlamaConfig", "DoRALinear", "LoRALinear", "apply_linear_adapter_"]
