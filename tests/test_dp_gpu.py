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
