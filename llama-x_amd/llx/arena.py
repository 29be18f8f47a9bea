"""Flat arena for the small trainable parameters (LoRA factors, DoRA magnitudes, norm weights, biases).

A LoRA fine-tune of Llama-3.1-8B has ~500 trainable tensors of 16-230 K elements each.  Left as separate tensors they cost, per step,
15 under-filled launches of the fused AdamW (each tensor is its own 64 K-element chunk list: ~40 of 256 CUs busy, 0.8 ms for 84 MB) and,
with more than one replica, one accumulate-add kernel per tensor to get the gradients into the flat exchange buckets (1.5 ms).

The arena lays them out ONCE, per dtype, in one parameter buffer P and one gradient buffer G of identical layout:

    param.data   = view of P            (the model computes from P; nothing else changes for the modules)
    param._llx_slot = (G, offset, numel) (where backward may write this parameter's gradient directly: llx.ops._grad_dst)

* the fused blocks' backward writes the adapter gradients straight into G (the reduce stage of the weight-gradient products takes the
  arena slice as its output: the members of a fused group sit back to back in the order the kernel emits them), autograd then adopts
  the returned view as ``param.grad`` - no copy, no accumulate kernel;
* the optimizer sees ONE parameter per dtype (``arena.params()``; AdamW is elementwise, so the update is bit-identical to the
  per-tensor one) - one full-width launch;
* a data-parallel bucket is a slice of G (llx.dp.GradBuckets picks the arena up from the model): zero-copy exchange.

Dense trainable weights (embeddings, LM head, conv kernels: ndim >= 2 and not an adapter factor) stay ordinary parameters: they are big
enough to fill the chip on their own, and the host-side weight images (transposed / concatenated copies, llx.ops._cached) are keyed on
their version counters, which an update through a flat alias would not bump.

Accumulation micro-steps and anything that does not write in place stay correct: a gradient that finds ``param.grad`` already set (or
whose slot another autograd node of the same backward has already claimed: a shared norm / adapter) is produced in a fresh buffer and
added by autograd, and ``settle()`` repairs whatever ended up elsewhere (copies a foreign ``.grad`` into its slot, zeroes the slot of a
parameter that got no gradient) before the exchange / optimizer read G.

The arena is self-contained for the reference's plain loop ``loss.backward(); optim.step(); optim.zero_grad()``
(train_metamathqa.py:253-254, train_librispeech.py:243-244): it registers optimizer step hooks that fire for any optimizer built on
``arena.params()``.  Before the step the flat gradient is re-attached (``zero_grad(set_to_none=True)`` drops it), ``settle()`` runs, and
members WITHOUT a gradient this step are put aside; after the step their parameter values are restored (the per-tensor optimizer of the
reference skips such parameters entirely - no weight decay, no moment decay; with zero moments the flat update is exactly the decay
term, which the restore undoes) and every member's ``.grad`` view is dropped so that the next backward writes G in place again instead of
accumulating.  One stated difference remains: the flat optimizer has ONE step counter, so a member that sits idle for some steps and then
receives gradients sees the bias correction of the global step, not of its own count.
Build the arena LAST - after adapters are applied, the base is quantised, requires_grad flags are final and the model sits on its device
(``model.to(...)`` re-creates parameter storage and would cut the views; ``load_state_dict`` copies in place and is fine).  ``verify()``
checks that every trainable parameter is still covered and still aliases the arena.
The checkpoint wire format keeps the reference's per-parameter optimizer state (train_metamathqa.py:259-265), numbered as the
reference numbers it: its optimizer is built on ``model.parameters()`` (train_metamathqa.py:188, train_librispeech.py:181), frozen
parameters included - they take an index and simply have no state entry.  ``optim_state_dict`` / ``load_optim_state_dict`` translate
(shapes checked entry by entry).
"""
from __future__ import annotations

import weakref
from typing import Iterable, Optional

import torch
from torch import Tensor, nn

_ALIGN = 128  # elements: every slot starts 256-byte aligned (the kernels read parameters with 16-byte vector loads)


def _is_small(name: str, p: Tensor) -> bool:
    return p.ndim == 1 or name.rsplit(".", 1)[-1] in ("lora_a", "lora_b")


def _layer_order(layer: nn.Module) -> list[nn.Parameter]:
    """Parameters of one transformer layer in the order the fused groups emit their gradients: per group (q|k|v, o, gate|up, down) the
    B factors of its members back to back, then the A factors; everything else of the layer in registration order after them."""
    out: list[nn.Parameter] = []
    seen: set[int] = set()

    def add(p):
        if p is not None and isinstance(p, nn.Parameter) and id(p) not in seen:
            seen.add(id(p))
            out.append(p)

    for sub in layer.modules():
        plans = getattr(sub, "plans", None)
        if plans is None:
            continue
        for g in plans():
            members = [m for m in g.members if getattr(m, "rank", 0) > 0]
            for m in members:
                add(m.lora_b)
            for m in members:
                add(m.lora_a)
    for p in layer.parameters():
        add(p)
    return out


class TrainableArena:
    def __init__(self, model: nn.Module):
        self.model_order = list(model.parameters())  # = the parameter numbering of optim_cls(model.parameters()) (checkpoint wire format)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        small = {id(p) for n, p in named if _is_small(n, p)}
        layer_of: dict[int, nn.Module] = {}
        for layer in getattr(model, "layers", []):
            for p in layer.parameters():
                layer_of[id(p)] = layer
        ordered: list[nn.Parameter] = []
        done: set[int] = set()
        for _, p in named:
            if id(p) in done or id(p) not in small:
                continue
            group = _layer_order(layer_of[id(p)]) if id(p) in layer_of else [p]
            for q in group:
                if id(q) in small and id(q) not in done and q.requires_grad:
                    done.add(id(q))
                    ordered.append(q)
        self.members = ordered
        self.dense = [p for _, p in named if id(p) not in small]
        self.flat: list[nn.Parameter] = []  # one per (dtype, device)
        self._G: list[Tensor] = []          # the gradient buffer of each flat parameter (flat.grad may be dropped by zero_grad)
        self._idle: list[tuple[nn.Parameter, Tensor]] = []  # members without a gradient this step: (member, parameter values to restore)
        self._slot_of: dict[int, tuple[int, int, int]] = {}  # id(param) -> (flat index, offset, numel)
        by_key: dict = {}
        for p in ordered:
            by_key.setdefault((p.dtype, p.device), []).append(p)
        for fi, ((dtype, device), ps) in enumerate(by_key.items()):
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += -(-p.numel() // _ALIGN) * _ALIGN
            P = torch.zeros(total, dtype=dtype, device=device)
            G = torch.zeros(total, dtype=dtype, device=device)
            with torch.no_grad():
                for p, o in zip(ps, offs):
                    P[o : o + p.numel()].copy_(p.detach().reshape(-1))
                    p.data = P[o : o + p.numel()].view(p.shape)
                    p._llx_slot = (G, o, p.numel())
                    self._slot_of[id(p)] = (fi, o, p.numel())
            fp = nn.Parameter(P, requires_grad=True)
            fp.grad = G
            self.flat.append(fp)
            self._G.append(G)
        model._llx_arena = self
        self._install_step_hooks()

    # ---- optimizer step hooks: what makes `loss.backward(); optim.step(); optim.zero_grad()` correct without further calls
    def _owns(self, optim: torch.optim.Optimizer) -> bool:
        return bool(self.flat) and any(p is self.flat[0] for g in optim.param_groups for p in g["params"])

    def _install_step_hooks(self):
        from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook

        ref = weakref.ref(self)
        handles: list = []

        def pre(optim, args, kwargs):
            arena = ref()
            if arena is not None and arena._owns(optim):
                arena.before_step()

        def post(optim, args, kwargs):
            arena = ref()
            if arena is not None and arena._owns(optim):
                arena.after_step()

        handles += [register_optimizer_step_pre_hook(pre), register_optimizer_step_post_hook(post)]
        self._hook_handles = handles
        weakref.finalize(self, lambda hs=handles: [h.remove() for h in hs])  # the global hooks go when the arena goes

    def close(self):
        """Remove the optimizer step hooks (the views stay)."""
        for h in getattr(self, "_hook_handles", []):
            h.remove()
        self._hook_handles = []

    def before_step(self):
        for fp, G in zip(self.flat, self._G):
            if fp.grad is not G:
                fp.grad = G  # optim.zero_grad(set_to_none=True) dropped it: without this the optimizer would skip the whole arena
        self._idle = [(p, p.detach().clone()) for p in self.members if p.grad is None]
        self.settle()

    def after_step(self):
        with torch.no_grad():
            for p, keep in self._idle:
                p.copy_(keep)  # a member without a gradient is skipped by the reference's per-tensor optimizer
        self._idle = []
        for p in self.members:
            p.grad = None  # the next backward writes its slot of G in place (a kept view would make autograd accumulate)

    # ---- what the optimizer is built on
    def params(self) -> list[nn.Parameter]:
        return self.flat + self.dense

    def verify(self, model: nn.Module) -> None:
        """Raises if the model's trainable parameters are no longer exactly the arena's (an adapter added, a flag flipped, the model moved
        to another device after the arena was built): the flat optimizer would silently miss or mis-address them."""
        mine = {id(p) for p in self.members} | {id(p) for p in self.dense}
        now = {id(p) for p in model.parameters() if p.requires_grad}
        if mine != now:
            raise RuntimeError(f"TrainableArena is stale: {len(now - mine)} trainable parameter(s) not covered, {len(mine - now)} no longer trainable; "
                               "build the arena after the model is final")
        for p in self.members:
            fi, o, n = self._slot_of[id(p)]
            P = self.flat[fi]
            if p.device != P.device or p.data_ptr() != P.data_ptr() + o * P.element_size():
                raise RuntimeError("TrainableArena is stale: a member parameter no longer aliases the arena (was the model moved or re-materialised?)")

    def grad_view(self, p: nn.Parameter) -> Tensor:
        G, o, n = p._llx_slot
        return G[o : o + n].view(p.shape)

    def contains(self, p: nn.Parameter) -> bool:
        return id(p) in self._slot_of

    def ranges(self, params: Iterable[nn.Parameter]) -> list[Tensor]:
        """The slices of G that cover `params` (arena members), merged where they are adjacent: a bucket of the gradient exchange."""
        spans = sorted((self._slot_of[id(p)][0], self._slot_of[id(p)][1], -(-self._slot_of[id(p)][2] // _ALIGN) * _ALIGN) for p in params)
        out: list[list[int]] = []
        for fi, o, n in spans:
            if out and out[-1][0] == fi and out[-1][2] == o:
                out[-1][2] = o + n
            else:
                out.append([fi, o, o + n])
        return [self._G[fi][lo:hi] for fi, lo, hi in out]

    def settle(self, params: Optional[Iterable[nn.Parameter]] = None):
        """Make G hold this step's gradient of every given member (default: all) before something reads G as a whole: a ``.grad`` that
        lives elsewhere is copied into its slot and re-pointed, a missing one zeroes its slot (the flat optimizer would otherwise
        apply last step's values; zero gradient = the per-tensor optimizer's skip, up to moment decay)."""
        for p in self.members if params is None else params:
            if id(p) not in self._slot_of:
                continue
            p._llx_claimed = False
            dst = self.grad_view(p)
            if p.grad is None:
                dst.zero_()
            elif p.grad.data_ptr() != dst.data_ptr():
                dst.copy_(p.grad)
                p.grad = dst

    def zero_grad(self):
        """Drop the per-parameter views (the next backward writes G in place); G itself needs no memset."""
        for p in self.members:
            p.grad = None
            p._llx_claimed = False
        for p in self.dense:
            p.grad = None

    # ---- checkpoint wire format: per-parameter optimizer state, numbered as optim_cls(model.parameters()) numbers it
    def optim_state_dict(self, optim: torch.optim.Optimizer) -> dict:
        sd = optim.state_dict()
        n_flat = len(self.flat)
        assert len(sd["param_groups"]) == 1 and sd["param_groups"][0]["params"] == list(range(n_flat + len(self.dense))), \
            "arena translation expects one parameter group built from arena.params()"
        index = {id(p): i for i, p in enumerate(self.model_order)}
        state: dict = {}
        for p in self.members:
            fi, o, n = self._slot_of[id(p)]
            st = sd["state"].get(fi)
            if st is None:
                continue
            state[index[id(p)]] = {k: (v[o : o + n].view(p.shape).clone() if torch.is_tensor(v) and v.dim() > 0 else (v.clone() if torch.is_tensor(v) else v))
                                   for k, v in st.items()}
        for j, p in enumerate(self.dense):
            if n_flat + j in sd["state"]:
                state[index[id(p)]] = sd["state"][n_flat + j]
        group = dict(sd["param_groups"][0], params=list(range(len(self.model_order))))
        return {"state": dict(sorted(state.items())), "param_groups": [group]}

    def load_optim_state_dict(self, optim: torch.optim.Optimizer, sd: dict):
        index = {id(p): i for i, p in enumerate(self.model_order)}
        n_flat = len(self.flat)
        listed = [i for g in sd["param_groups"] for i in g["params"]]
        if len(listed) != len(self.model_order):
            raise ValueError(f"optimizer checkpoint numbers {len(listed)} parameters, the model has {len(self.model_order)} "
                             "(the reference builds its optimizer on model.parameters(), frozen ones included)")
        for p in self.model_order:
            st = sd["state"].get(index[id(p)])
            if st is None:
                continue
            if not p.requires_grad:
                raise ValueError(f"optimizer checkpoint holds state for parameter {index[id(p)]}, which is frozen in this model")
            for k, v in st.items():
                if torch.is_tensor(v) and v.dim() > 0 and tuple(v.shape) != tuple(p.shape):
                    raise ValueError(f"optimizer checkpoint entry {index[id(p)]}[{k}] has shape {tuple(v.shape)}, the parameter {tuple(p.shape)}")
        state: dict = {}
        for fi, fp in enumerate(self.flat):
            mine = [p for p in self.members if self._slot_of[id(p)][0] == fi and index[id(p)] in sd["state"]]
            if not mine:
                continue
            proto = sd["state"][index[id(mine[0])]]
            st = {}
            for k, v in proto.items():
                if torch.is_tensor(v) and v.dim() > 0:
                    buf = torch.zeros(fp.numel(), dtype=v.dtype, device=fp.device)
                    for p in mine:
                        _, o, n = self._slot_of[id(p)]
                        buf[o : o + n].copy_(sd["state"][index[id(p)]][k].reshape(-1))
                    st[k] = buf
                else:
                    st[k] = v
            state[fi] = st
        for j, p in enumerate(self.dense):
            if index[id(p)] in sd["state"]:
                state[n_flat + j] = sd["state"][index[id(p)]]
        group = dict(sd["param_groups"][0], params=list(range(n_flat + len(self.dense))))
        optim.load_state_dict({"state": state, "param_groups": [group]})
