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
    opt_ref = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], **kw)
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
            assert n == "unused" or torch.allclose(p, q, rtol=1e-6, atol=1e-7), (step, n)
    # `unused` never had a gradient: the per-tensor optimizer skipped it; the arena zeroed its slot, and a zero gradient with zero
    # moments moves nothing but the weight decay - which the reference optimizer did not apply.  That is the one stated difference:
    assert torch.equal(ref.unused, torch.ones(7)) and torch.allclose(m.unused, torch.ones(7) * (1 - 1e-2 * 0.01) ** 3)
    # checkpoint: per-parameter numbering of torch.optim.AdamW(trainables in model order)
    sd, sd_ref = tr.state_dict()["optim"], opt_ref.state_dict()
    assert sd["param_groups"][0]["params"] == sd_ref["param_groups"][0]["params"]
    names = [n for n, p in ref.named_parameters() if p.requires_grad]
    for i, n in enumerate(names):
        if n == "unused":
            assert i not in sd_ref["state"]  # (the arena reports zero moments for it)
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
