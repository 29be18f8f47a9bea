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

Accumulation micro-steps and anything that does not write in place stay correct: a gradient that finds ``param.grad`` already set is
produced in a fresh buffer and added by autograd (into the arena view), and ``settle()`` repairs whatever ended up elsewhere (copies a
foreign ``.grad`` into its slot, zeroes the slot of a parameter that got no gradient) before the exchange / optimizer read G.
Build the arena LAST - after adapters are applied, the base is quantised, requires_grad flags are final and the model sits on its device
(``model.to(...)`` re-creates parameter storage and would cut the views; ``load_state_dict`` copies in place and is fine).  ``verify()``
checks that every trainable parameter is still covered and still aliases the arena.
The checkpoint wire format keeps the reference's per-parameter optimizer state (train_metamathqa.py:259-265): ``optim_state_dict`` /
``load_optim_state_dict`` translate.
"""
from __future__ import annotations

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
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.model_order = [p for _, p in named]  # = the parameter numbering of torch.optim.X(trainables) (checkpoint wire format)
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
        model._llx_arena = self

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
        return [self.flat[fi].grad[lo:hi] for fi, lo, hi in out]

    def settle(self, params: Optional[Iterable[nn.Parameter]] = None):
        """Make G hold this step's gradient of every given member (default: all) before something reads G as a whole: a ``.grad`` that
        lives elsewhere is copied into its slot and re-pointed, a missing one zeroes its slot (the flat optimizer would otherwise
        apply last step's values; zero gradient = the per-tensor optimizer's skip, up to moment decay)."""
        for p in self.members if params is None else params:
            if id(p) not in self._slot_of:
                continue
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
        for p in self.dense:
            p.grad = None

    # ---- checkpoint wire format: per-parameter optimizer state, numbered as torch.optim.X(trainable parameters in model order)
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
