"""llx.arena.TrainableArena on the CPU (host logic only: the in-place gradient writes of the HIP backward are covered by the gpu tests):
parameters become views of one flat buffer, gradients produced elsewhere are gathered by settle(), the flat optimizer gives the same
parameters as the per-tensor one, and the optimizer state translates to / from the reference's per-parameter checkpoint format."""
import copy
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))


class Toy(nn.Module):
    """A frozen dense weight with a LoRA pair, a norm-like 1-D weight, and one dense TRAINABLE matrix (stays outside the arena)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.weight = nn.Parameter(torch.randn(24, 16, generator=g), requires_grad=False)
        self.lora_a = nn.Parameter(torch.randn(4, 16, generator=g) * 0.1)
        self.lora_b = nn.Parameter(torch.randn(24, 4, generator=g) * 0.1)
        self.gain = nn.Parameter(torch.ones(24))
        self.unused = nn.Parameter(torch.ones(7))  # never gets a gradient
        self.head = nn.Linear(24, 3)

    def forward(self, x):
        h = x @ self.weight.T + (x @ self.lora_a.T) @ self.lora_b.T
        return self.head(torch.tanh(h) * self.gain).square().sum()


def _data(i):
    return torch.randn(5, 16, generator=torch.Generator().manual_seed(100 + i))


def test_arena_layout_and_views():
    from llx.arena import TrainableArena

    m = Toy()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    arena = TrainableArena(m)
    assert [p is q for p, q in zip(arena.members, (m.lora_a, m.lora_b, m.gain, m.unused, m.head.bias))] == [True] * 5
    assert arena.dense == [m.head.weight] and len(arena.flat) == 1 and arena.params()[0] is arena.flat[0]
    P = arena.flat[0]
    for n, p in m.named_parameters():
        assert torch.equal(p, before[n])  # values survive the move
        if arena.contains(p):
            assert P.data_ptr() <= p.data_ptr() < P.data_ptr() + P.numel() * 4 and p.data_ptr() % 256 == P.data_ptr() % 256
            assert arena.grad_view(p).shape == p.shape
    with torch.no_grad():
        P.add_(1.0)  # an update of the flat parameter IS an update of the members
    assert torch.equal(m.lora_a, before["lora_a"] + 1.0)
    arena.verify(m)
    m.weight.requires_grad_(True)  # a parameter turned trainable behind the arena's back
    try:
        arena.verify(m)
        raise AssertionError("verify() must notice the uncovered parameter")
    except RuntimeError as e:
        assert "stale" in str(e)
    m.weight.requires_grad_(False)
    # buckets of the exchange: adjacent members merge into one slice
    assert len(arena.ranges([m.lora_a, m.lora_b, m.gain])) == 1 and len(arena.ranges([m.lora_a, m.gain])) == 2


def test_flat_optimizer_matches_per_tensor_optimizer_and_checkpoint_format(tmp_path):
    from llx.arena import TrainableArena
    from llx.train import Trainer

    ref = Toy()
    m = copy.deepcopy(ref)
    arena = TrainableArena(m)
    kw = dict(lr=1e-2, weight_decay=0.01, betas=(0.9, 0.95))
    opt_ref = torch.optim.AdamW(ref.parameters(), **kw)  # as the reference scripts build it: frozen parameters included (train_metamathqa.py:188)
    opt = torch.optim.AdamW(arena.params(), **kw)
    tr = Trainer(m, opt, grad_accum=2, clip_grad_norm=0.5)
    for step in range(3):
        xs = [_data(2 * step), _data(2 * step + 1)]
        for x in xs:
            (ref(x) / 2).backward()
        torch.nn.utils.clip_grad_norm_([p for p in ref.parameters() if p.requires_grad], 0.5)
        opt_ref.step()
        opt_ref.zero_grad()
        tr.step([lambda mm, x=x: mm(x) for x in xs])
        for (n, p), (_, q) in zip(ref.named_parameters(), m.named_parameters()):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (step, n)
    # `unused` never had a gradient: the per-tensor optimizer skipped it (no weight decay either); the arena's step hooks put its
    # values aside and restore them, so it does not drift
    assert torch.equal(ref.unused, torch.ones(7)) and torch.equal(m.unused, torch.ones(7))
    # checkpoint: per-parameter numbering of torch.optim.AdamW(model.parameters()) - the frozen `weight` takes index 0 and has no state
    sd, sd_ref = tr.state_dict()["optim"], opt_ref.state_dict()
    assert sd["param_groups"][0]["params"] == sd_ref["param_groups"][0]["params"] == list(range(7))
    names = [n for n, p in ref.named_parameters()]
    assert names[0] == "weight" and 0 not in sd["state"] and 0 not in sd_ref["state"]
    assert sorted(k for k in sd["state"] if names[k] != "unused") == sorted(sd_ref["state"])
    for i, n in enumerate(names):
        if n in ("unused", "weight"):
            assert i not in sd_ref["state"]  # (the arena reports zero moments for `unused`)
            continue
        for k in ("exp_avg", "exp_avg_sq"):
            assert sd["state"][i][k].shape == sd_ref["state"][i][k].shape, (n, k)
            assert torch.allclose(sd["state"][i][k], sd_ref["state"][i][k], rtol=1e-5, atol=1e-8), (n, k)
        assert float(sd["state"][i]["step"]) == float(sd_ref["state"][i]["step"]) == 3.0
    # resume: a fresh arena model + optimizer loaded from the per-parameter dict continues exactly like the original
    m2 = copy.deepcopy(Toy())
    arena2 = TrainableArena(m2)
    opt2 = torch.optim.AdamW(arena2.params(), **kw)
    tr2 = Trainer(m2, opt2, grad_accum=2, clip_grad_norm=0.5)
    # through the file format of the reference scripts: torch.save(dict(step, model, optim)) / torch.load(weights_only=True)
    # (train_metamathqa.py:259-265, train_librispeech.py:200-204); the members are views of one storage, which is written once
    torch.save(tr.state_dict(), tmp_path / "last.pth")
    assert (tmp_path / "last.pth").stat().st_size < 40_000
    tr2.load_state_dict(torch.load(tmp_path / "last.pth", weights_only=True))
    xs = [_data(10), _data(11)]
    tr.step([lambda mm, x=x: mm(x) for x in xs])
    tr2.step([lambda mm, x=x: mm(x) for x in xs])
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q), n
    # a checkpoint written by the REFERENCE's optimizer (AdamW over model.parameters()) loads into the arena and continues identically
    m3 = Toy()
    m3.load_state_dict(ref.state_dict())
    arena3 = TrainableArena(m3)
    opt3 = torch.optim.AdamW(arena3.params(), **kw)
    arena3.load_optim_state_dict(opt3, copy.deepcopy(opt_ref.state_dict()))
    x = _data(20)
    ref(x).backward()
    opt_ref.step()
    opt_ref.zero_grad()
    m3(x).backward()
    opt3.step()
    opt3.zero_grad()
    for (n, p), (_, q) in zip(ref.named_parameters(), m3.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), n
    # ... and a checkpoint numbered over the trainables only (or with the wrong shapes) is refused instead of mis-assigned
    bad = copy.deepcopy(opt_ref.state_dict())
    bad["param_groups"][0]["params"] = list(range(6))
    try:
        arena3.load_optim_state_dict(opt3, bad)
        raise AssertionError("a checkpoint with another parameter numbering must be refused")
    except ValueError as e:
        assert "numbers 6 parameters" in str(e)
    bad = copy.deepcopy(opt_ref.state_dict())
    bad["state"][1]["exp_avg"] = bad["state"][2]["exp_avg"]  # lora_a's moment replaced by lora_b's
    try:
        arena3.load_optim_state_dict(opt3, bad)
        raise AssertionError("a mis-shaped state entry must be refused")
    except ValueError as e:
        assert "shape" in str(e)


def test_plain_reference_loop_without_trainer():
    """The reference's loop body (train_metamathqa.py:226-254): loss.backward(); [clip]; optim.step(); optim.zero_grad() - no Trainer, no
    GradBuckets, no explicit settle(): the arena's optimizer step hooks keep it correct (zero_grad(set_to_none=True) drops the flat
    gradient, CPU autograd produces gradients outside the arena, members' .grad must not accumulate across steps)."""
    from llx.arena import TrainableArena

    ref = Toy()
    m = copy.deepcopy(ref)
    kw = dict(lr=1e-2, weight_decay=0.05)
    opt_ref = torch.optim.AdamW(ref.parameters(), **kw)
    opt = torch.optim.AdamW(TrainableArena(m).params(), **kw)  # the one changed line
    for step in range(4):
        for mm, oo in ((ref, opt_ref), (m, opt)):
            for micro in range(2):  # gradient accumulation as the scripts do it
                (mm(_data(2 * step + micro)) / 2).backward()
            torch.nn.utils.clip_grad_norm_(mm.parameters(), 0.7)
            oo.step()
            oo.zero_grad()
        for (n, p), (_, q) in zip(ref.named_parameters(), m.named_parameters()):
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (step, n)
        assert all(p.grad is None for p in m.parameters())
    assert torch.equal(m.unused, torch.ones(7))


def test_grad_slot_is_claimed_once_per_backward():
    """llx.ops._grad_dst hands a parameter's arena slot to ONE producer per backward: a second autograd node of the same backward (a
    shared norm weight / adapter) gets None, writes a fresh buffer and autograd sums; the claim is released by settle()."""
    from llx.arena import TrainableArena
    from llx.ops import _grad_dst

    m = Toy()
    arena = TrainableArena(m)
    first = _grad_dst([m.gain])
    assert first is not None and first.data_ptr() == arena.grad_view(m.gain).data_ptr()
    assert _grad_dst([m.gain]) is None, "second producer in the same backward must not get the same slice"
    assert _grad_dst([m.lora_a, m.lora_b]) is None, "slots padded apart (64 of 128 elements used): not one contiguous slice"
    assert _grad_dst([m.lora_b]) is not None and _grad_dst([m.lora_b]) is None
    arena.settle()
    assert _grad_dst([m.gain]) is not None
    arena.zero_grad()
    m.gain.grad = torch.ones(24)
    assert _grad_dst([m.gain]) is None, "accumulation micro-step: autograd has to add"


def test_settle_gathers_foreign_gradients_and_zeroes_missing_ones():
    from llx.arena import TrainableArena

    m = Toy()
    arena = TrainableArena(m)
    G = arena.flat[0].grad
    G.fill_(7.0)  # stale values of a previous step
    m(_data(0)).backward()
    assert m.unused.grad is None and m.lora_a.grad.data_ptr() != arena.grad_view(m.lora_a).data_ptr()  # CPU autograd wrote elsewhere
    want = m.lora_b.grad.clone()
    arena.settle()
    assert m.lora_b.grad.data_ptr() == arena.grad_view(m.lora_b).data_ptr() and torch.equal(arena.grad_view(m.lora_b), want)
    assert torch.equal(arena.grad_view(m.unused), torch.zeros(7))
    arena.zero_grad()
    assert all(p.grad is None for p in m.parameters())
