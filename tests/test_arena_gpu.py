"""GPU: the trainable arena (llx/arena.py) on the real backward - the fused blocks write the LoRA / norm gradients straight into the
arena's gradient buffer, the flat fused AdamW gives bit-identical parameters to the per-tensor one (eager, whole-step hipGraph,
accumulation micro-steps, staged data-parallel step over RCCL)."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from oracle import ref as O  # noqa: E402
from tests.util import bf16_params, build_model  # noqa: E402


def _setup(cuda, n_layers=3, kv_heads=1, quantize=None):
    cfg = O.TINY._replace(num_layers=n_layers, num_kv_heads=kv_heads)
    p = O.init_params(cfg)
    p.update(O.init_lora(cfg, 8))
    pb, _ = bf16_params(p)
    batches = []
    for s in range(3):
        t = O.randint("tokens", (1, 256), 0, cfg.vocab_size, s).to(cuda)
        batches.append((t, torch.roll(t, -1, 1)))

    def make():
        model = build_model(cfg, pb, cuda, lora_rank=8, quantize=quantize, quantize_kwargs=dict(dynamic_int8_act=True) if quantize else None)
        for n, q in model.named_parameters():
            q.requires_grad_("lora_" in n or n.endswith("norm.weight"))
        return model

    return cfg, make, batches


def _adamw(params, capturable=False):
    return torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01, fused=True, capturable=capturable)


def test_backward_writes_gradients_in_place(cuda):
    from llx.arena import TrainableArena

    # two kv heads: the q|k|v boundaries (512, 768, 1024) are multiples of 256 as at 8B dimensions, so the B factors take the segmented
    # in-place route too (with TINY's single kv head they fall back to slicing copies, which settle() gathers: the other tests)
    _, make, batches = _setup(cuda, kv_heads=2)
    ref = make()
    ref(batches[0][0], labels=batches[0][1]).backward()
    model = make()
    arena = TrainableArena(model)
    assert len(arena.flat) == 1 and not arena.dense and len(arena.members) == sum(1 for p in model.parameters() if p.requires_grad)
    # the members of every fused group are back to back: q|k|v B factors, then A factors, ...
    lay = model.layers[0].attention
    gv = arena.grad_view
    assert gv(lay.wk.lora_b).data_ptr() == gv(lay.wq.lora_b).data_ptr() + lay.wq.lora_b.numel() * 2
    assert gv(lay.wv.lora_a).data_ptr() == gv(lay.wk.lora_a).data_ptr() + lay.wk.lora_a.numel() * 2
    arena.flat[0].grad.fill_(float("nan"))  # whatever is not written this step would show
    model(batches[0][0], labels=batches[0][1]).backward()
    torch.cuda.synchronize()
    for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        if p.requires_grad:
            assert p.grad.data_ptr() == gv(p).data_ptr(), f"{n}: gradient did not land in the arena"
            assert torch.equal(p.grad, r.grad), n
    assert not torch.isnan(arena.flat[0].grad).any()  # (the layout has no padding gaps here: every slot is a multiple of 128 elements)
    # second backward WITHOUT zero_grad = an accumulation micro-step: fresh buffers, autograd adds into the arena views
    ref(batches[1][0], labels=batches[1][1]).backward()
    model(batches[1][0], labels=batches[1][1]).backward()
    for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        if p.requires_grad:
            assert p.grad.data_ptr() == gv(p).data_ptr() and torch.equal(p.grad, r.grad), n


@pytest.mark.parametrize("mode", ["eager", "graph", "eager-int8"])
def test_flat_adamw_step_is_bit_identical(cuda, mode):
    from llx.arena import TrainableArena
    from llx.dp import GradBuckets

    # "-int8": INT8 base with dynamically quantised activations + bf16 LoRA (BASELINE configs[3]); two kv heads = the in-place route for all
    _, make, batches = _setup(cuda, kv_heads=2, quantize="int8") if mode.endswith("int8") else _setup(cuda)
    mode = mode.split("-")[0]
    ref = make()
    opt_ref = _adamw([p for p in ref.parameters() if p.requires_grad])
    ref_losses = []
    for t, l in batches:
        loss = ref(t, labels=l)
        loss.backward()
        opt_ref.step()
        opt_ref.zero_grad()
        ref_losses.append(loss.item())
    model = make()
    arena = TrainableArena(model)
    opt = _adamw(arena.params(), capturable=mode == "graph")
    assert len(opt.param_groups[0]["params"]) == 1
    buckets = GradBuckets(model)  # single replica: inactive, but finish() settles the arena and zero_grad() drops the views
    losses = []
    if mode == "eager":
        for t, l in batches:
            loss = model(t, labels=l)
            loss.backward()
            buckets.finish()
            opt.step()
            buckets.zero_grad()
            losses.append(loss.item())
    else:
        tok, lab = batches[0][0].clone(), batches[0][1].clone()
        snap = arena.flat[0].detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                buckets.zero_grad()
                model(tok, labels=lab).backward()
                buckets.finish()
                opt.step()
        torch.cuda.current_stream().wait_stream(side)
        buckets.zero_grad()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_loss = model(tok, labels=lab)
            static_loss.backward()
            buckets.finish()
            opt.step()
        with torch.no_grad():  # undo the warm-up / capture-time updates
            arena.flat[0].copy_(snap)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        for t, l in batches:
            tok.copy_(t)
            lab.copy_(l)
            g.replay()
            losses.append(static_loss.item())
    torch.cuda.synchronize()
    assert losses == ref_losses, (losses, ref_losses)
    for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
        if p.requires_grad:  # (the frozen Int8LinearWeight base has no aten.equal, as in the reference)
            assert torch.equal(p, r), n


def test_staged_step_on_arena_buckets(cuda):
    """StagedStep over RCCL (world of one) with the arena: one bucket per stage = one slice of the arena's gradient buffer."""
    from llx.arena import TrainableArena
    from llx.dp import StagedStep, llama_stages

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29579")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        _, make, batches = _setup(cuda)
        ref = make()
        opt_ref = _adamw([p for p in ref.parameters() if p.requires_grad])
        ref_losses = []
        for t, l in batches[:2]:
            loss = ref(t, labels=l)
            loss.backward()
            opt_ref.step()
            opt_ref.zero_grad()
            ref_losses.append(loss.item())
        for mode in ("eager", "graph"):
            model = make()
            arena = TrainableArena(model)
            opt = _adamw(arena.params(), capturable=True)
            tok, lab = batches[0][0].clone(), batches[0][1].clone()
            stages, sp = llama_stages(model, 3, labels=lab)
            stepper = StagedStep(model, stages, sp, opt, graph=mode == "graph", force=True)
            bs = stepper.buckets.buckets
            assert len(bs) == 3 and all(b["arena"] and len(b["flats"]) == 1 for b in bs)
            G = arena.flat[0].grad
            assert sorted(b["flat"].data_ptr() for b in bs)[0] == G.data_ptr() and sum(b["flat"].numel() for b in bs) == G.numel()
            if mode == "graph":
                snap = arena.flat[0].detach().clone()
                stepper.capture(tok)
                with torch.no_grad():
                    arena.flat[0].copy_(snap)
                    for st in opt.state.values():
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
            losses = []
            for t, l in batches[:2]:
                tok.copy_(t)
                lab.copy_(l)
                losses.append(float(stepper(tok)))
            torch.cuda.synchronize()
            assert losses == ref_losses, (mode, losses, ref_losses)
            for (n, p), (_, r) in zip(model.named_parameters(), ref.named_parameters()):
                assert torch.equal(p, r), (mode, n)
    finally:
        dist.destroy_process_group()
