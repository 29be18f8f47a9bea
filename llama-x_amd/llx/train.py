"""Training-step semantics of the reference loop (train_metamathqa.py:217-257 / train_librispeech.py:212-247) plus the
data-parallel exchange, as a reusable harness:

    for micro-batch in accumulation:  loss = model(...); (loss / accum).backward()      # exchange only on the last one
    lr_schedule.set_lr(optim, step)   # LR is set BEFORE the optimizer step
    [clip_grad_norm_]
    optim.step(); zero_grad

and the checkpoint dict {step, model, optim} of :259-265 / resume of train_librispeech.py:200-204.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch
from torch import nn

from .data import LRScheduler
from .dp import GradBuckets


class Trainer:
    def __init__(self, model: nn.Module, optim: torch.optim.Optimizer, *, lr_schedule: Optional[LRScheduler] = None, grad_accum: int = 1,
                 clip_grad_norm: Optional[float] = None, n_buckets: int = 4):
        self.model, self.optim, self.lr_schedule = model, optim, lr_schedule
        self.grad_accum, self.clip = grad_accum, clip_grad_norm
        if self._arena() is not None:
            self._arena().verify(model)
        self.buckets = GradBuckets(model, n_buckets=n_buckets)
        self.step_idx = 0
        self.n_toks = torch.zeros((), dtype=torch.int64)

    def step(self, micro_batches: Iterable[Callable[[nn.Module], torch.Tensor]] | Callable[[nn.Module], torch.Tensor]) -> torch.Tensor:
        """``micro_batches``: one callable, or ``grad_accum`` callables, each mapping the model to its loss."""
        fns = [micro_batches] if callable(micro_batches) else list(micro_batches)
        assert len(fns) == self.grad_accum, (len(fns), self.grad_accum)
        loss = None
        for i, fn in enumerate(fns):
            self.buckets.sync_enabled = i == len(fns) - 1  # exchange gradients only after the last micro-batch
            loss = fn(self.model)
            (loss / self.grad_accum).backward()
        self.buckets.finish()
        if self.lr_schedule is not None:
            self.lr_schedule.set_lr(self.optim, self.step_idx)
        grad_norm = None
        if self.clip is not None:
            grad_norm = torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], self.clip)
        self.optim.step()
        self.buckets.zero_grad()
        self.step_idx += 1
        self.last_grad_norm = grad_norm
        return loss.detach()

    # ---- checkpoint wire format of the reference scripts
    def _arena(self):
        """The model's trainable arena if the optimizer was built on it (llx.arena.TrainableArena.params())."""
        arena = getattr(self.model, "_llx_arena", None)
        if arena is not None and arena.flat and any(p is arena.flat[0] for g in self.optim.param_groups for p in g["params"]):
            return arena
        return None

    def state_dict(self) -> dict:
        # the optimizer state keeps the reference's per-parameter numbering whether or not the parameters sit in an arena
        arena = self._arena()
        optim = arena.optim_state_dict(self.optim) if arena is not None else self.optim.state_dict()
        return dict(step=self.step_idx, model=self.model.state_dict(), optim=optim)

    def load_state_dict(self, ckpt: dict):
        self.step_idx = ckpt["step"]
        self.model.load_state_dict(ckpt["model"])
        arena = self._arena()
        if arena is not None:
            arena.load_optim_state_dict(self.optim, ckpt["optim"])
        else:
            self.optim.load_state_dict(ckpt["optim"])
