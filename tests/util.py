"""Shared helpers of the parity tests: build the product model from oracle-style parameter dicts."""
import torch

from oracle import ref as O


def bf16_params(p: dict) -> dict:
    """Round every floating tensor to bf16 (the GPU dtype) and return (bf16 dict, fp32 dict of the same rounded values)."""
    pb = {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in p.items()}
    pf = {k: (v.float() if v.is_floating_point() else v) for k, v in pb.items()}
    return pb, pf


def to_model_config(cfg: O.Cfg):
    from modelling import LlamaConfig

    return LlamaConfig(**{f: getattr(cfg, f) for f in LlamaConfig._fields})


def build_model(cfg: O.Cfg, params_bf16: dict, device, *, lora_rank: int = 0, lora_alpha: float | None = None, quantize: str | None = None,
                quantize_kwargs: dict | None = None, audio: bool = False):
    """Product model on ``device`` with the given weights; surgery order = quantise then adapt (train_metamathqa.py:178-179)."""
    from modelling import Llama, LlamaAudio, apply_linear_adapter_
    from subclasses import quantize_linear_

    model = (LlamaAudio if audio else Llama)(to_model_config(cfg))
    model = model.bfloat16()
    base = {k: v for k, v in params_bf16.items() if not (k.endswith(".lora_a") or k.endswith(".lora_b"))}
    missing = model.load_state_dict(base, strict=False)
    assert not missing.unexpected_keys, missing
    model.build_cache()
    if quantize:
        quantize_linear_(model.layers, quantize, **(quantize_kwargs or {}))
    if lora_rank:
        apply_linear_adapter_(model.layers, "lora", rank=lora_rank, alpha=float(lora_alpha if lora_alpha is not None else lora_rank))
        with torch.no_grad():
            for name, mod in model.layers.named_modules():
                key = f"layers.{name}"
                if key + ".lora_a" in params_bf16:
                    mod.lora_a.copy_(params_bf16[key + ".lora_a"])
                    mod.lora_b.copy_(params_bf16[key + ".lora_b"])
    return model.to(device)
