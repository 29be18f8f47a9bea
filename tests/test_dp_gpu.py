"""GPU: rehearsal of the N>1 gradient path on one MI355X - RCCL process group of size 1, flat bucket views, hook-launched and
finish()-launched exchanges give the same gradients as the plain single-replica backward (bitwise: one rank, sum of one)."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from oracle import ref as O  # noqa: E402
from tests.util import bf16_params, build_model  # noqa: E402


def test_grad_buckets_rccl_world1(cuda):
    from llx.dp import GradBuckets

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        cfg = O.TINY
        p = O.init_params(cfg)
        p.update(O.init_lora(cfg, 8))
        pb, _ = bf16_params(p)
        tokens = O.randint("tokens", (1, 256), 0, cfg.vocab_size).to(cuda)
        labels = torch.roll(tokens, -1, 1)

        def grads(mode):
            model = build_model(cfg, pb, cuda, lora_rank=8)
            for n, q in model.named_parameters():
                q.requires_grad_("lora_" in n or n.endswith("_norm.weight"))
            buckets = GradBuckets(model, n_buckets=3, force=mode != "plain", overlap=mode == "hooks")
            model(tokens, labels=labels).backward()
            buckets.finish()
            torch.cuda.synchronize()
            if mode != "plain":
                assert len(buckets.buckets) == 3
                assert all(q.grad.data_ptr() >= b["flat"].data_ptr() for b in buckets.buckets for q in b["params"])
            return {n: q.grad.clone() for n, q in model.named_parameters() if q.requires_grad}

        ref = grads("plain")
        for mode in ("hooks", "finish"):
            got = grads(mode)
            assert got.keys() == ref.keys()
            for n in ref:
                assert torch.equal(got[n], ref[n]), (mode, n)
    finally:
        dist.destroy_process_group()


def test_staged_step_matches_plain_step(cuda):
    """llx.dp.StagedStep (forward / backward in 3 stages, one flat bucket per stage handed to RCCL as soon as the stage's backward is
    queued; eager and replayed from per-stage hipGraphs) against the plain single-graph-free step: same loss and bit-identical
    parameters after two AdamW steps (RCCL world of one: sum of one, mean of one)."""
    from llx.dp import StagedStep, llama_stages

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        cfg = O.TINY._replace(num_layers=3)
        p = O.init_params(cfg)
        p.update(O.init_lora(cfg, 8))
        pb, _ = bf16_params(p)
        batches = []
        for s in range(2):
            t = O.randint("tokens", (1, 256), 0, cfg.vocab_size, s).to(cuda)
            batches.append((t, torch.roll(t, -1, 1)))

        def run(mode):
            model = build_model(cfg, pb, cuda, lora_rank=8)
            for n, q in model.named_parameters():
                q.requires_grad_("lora_" in n or n.endswith("norm.weight"))
            train = [q for q in model.parameters() if q.requires_grad]
            opt = torch.optim.AdamW(train, lr=1e-3, weight_decay=0.0, fused=True, capturable=True)
            losses = []
            if mode == "plain":
                for t, l in batches:
                    loss = model(t, labels=l)
                    loss.backward()
                    opt.step()
                    opt.zero_grad()
                    losses.append(loss.item())
            else:
                tok, lab = batches[0][0].clone(), batches[0][1].clone()
                stages, sp = llama_stages(model, 3, labels=lab)
                assert len(stages) == 3 and sum(len(x) for x in sp) == len(train)
                assert any(q is model.norm.weight for q in sp[-1]) and all(q is not model.norm.weight for q in sp[0])
                stepper = StagedStep(model, stages, sp, opt, graph=mode == "graph", force=True)
                assert len(stepper.buckets.buckets) == 3
                if mode == "graph":
                    snap = {n: q.detach().clone() for n, q in model.named_parameters()}
                    stepper.capture(tok)  # the warm-up steps inside capture() move the parameters: restore them
                    with torch.no_grad():
                        for n, q in model.named_parameters():
                            q.copy_(snap[n])
                    for st in opt.state.values():
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                v.zero_()
                for t, l in batches:
                    tok.copy_(t)
                    lab.copy_(l)
                    losses.append(float(stepper(tok)))
            torch.cuda.synchronize()
            return losses, {n: q.detach().clone() for n, q in model.named_parameters() if q.requires_grad}

        ref_l, ref_p = run("plain")
        for mode in ("eager", "graph"):
            got_l, got_p = run(mode)
            assert got_l == ref_l, (mode, got_l, ref_l)
            for n in ref_p:
                assert torch.equal(got_p[n], ref_p[n]), (mode, n)
    finally:
        dist.destroy_process_group()
