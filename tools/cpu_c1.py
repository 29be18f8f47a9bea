"""BASELINE.json configs[0] on the host CPU: "train_metamathqa.py: Llama-3.1-8B LoRA r=8, bs=1 seq=256, 10 steps on CPU eager PyTorch".

The reference script itself cannot run here (hub download, wandb, .cuda(), fused AdamW: SURVEY 8c); what runs is the oracle
(oracle/ref.py, the CPU restatement of the reference's forward pinned against it by oracle/gen_golden.py) driving the loop body of
train_metamathqa.py:217-257 - all 32 layers at the real dimensions, fp32 random-init weights (32 GB), LoRA r=8 alpha=8 on model.layers,
everything else frozen, AdamW lr 1e-4, B=1, S=256 with the first 64 positions unlabelled.  Prints one JSON object
(p50 step seconds, tokens/s, cores); run once per round on the GPU box's host and keep the output under profiles/.

    python tools/cpu_c1.py [--steps 10] [--layers 32]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from oracle import ref as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--seq", type=int, default=256)
    a = ap.parse_args()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 64)
    torch.set_num_threads(cores)
    cfg = O.LLAMA31_8B._replace(num_layers=a.layers, max_seq_len=a.seq)
    D, I, V = cfg.embed_dim, cfg.intermediate_dim, cfg.vocab_size
    hq, hkv = cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim
    g = torch.Generator().manual_seed(1234)
    t0 = time.perf_counter()
    # 8 G normal draws would take minutes on one core: one 64 M block of N(0, 0.02^2) values is drawn once and every weight is a (rolled)
    # copy of its head - timing and loss level do not care that tensors share values
    base = torch.randn(2**26, generator=g) * 0.02

    def rnd(o, n, k=[0]):
        k[0] += 1
        need = o * n
        src = base if need <= base.numel() else base.repeat(-(-need // base.numel()))
        return src.roll(7919 * k[0])[:need].view(o, n).clone()

    p = {"tok_embeddings.weight": rnd(V, D), "output.weight": rnd(V, D), "norm.weight": torch.ones(D)}
    shapes = {"attention.wq": (hq, D), "attention.wk": (hkv, D), "attention.wv": (hkv, D), "attention.wo": (D, hq),
              "feed_forward.w1": (I, D), "feed_forward.w3": (I, D), "feed_forward.w2": (D, I)}
    train = []
    for i in range(cfg.num_layers):
        pre = f"layers.{i}."
        p[pre + "attention_norm.weight"] = torch.ones(D)
        p[pre + "ffn_norm.weight"] = torch.ones(D)
        for suf, (o, n) in shapes.items():
            p[pre + suf + ".weight"] = rnd(o, n)
            p[pre + suf + ".lora_a"] = torch.randn(8, n, generator=g) * (0.5774 / n ** 0.5)  # kaiming_normal(a=sqrt 5), modelling/lora.py:34
            p[pre + suf + ".lora_b"] = torch.zeros(o, 8)                                      # modelling/lora.py:35
            train += [pre + suf + ".lora_a", pre + suf + ".lora_b"]
    init_s = time.perf_counter() - t0
    params = [p[k].requires_grad_(True) for k in train]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.0)
    times, losses = [], []
    for step in range(a.steps):
        tokens = torch.randint(0, V, (1, a.seq), generator=g)
        labels = torch.roll(tokens, -1, 1).clone()
        labels[:, :64] = -100
        labels[:, -1] = -100
        t0 = time.perf_counter()
        loss = O.llama_forward(tokens, p, cfg, labels=labels, lora_scale=1.0)
        loss.backward()
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t0)
        losses.append(float(loss.detach()))
        print(f"[cpu_c1] step {step}: {times[-1]:.2f} s, loss {losses[-1]:.4f}", file=sys.stderr, flush=True)
    ts = sorted(times[1:] or times)
    p50 = ts[len(ts) // 2]
    print(json.dumps({"config": "BASELINE.json configs[0]: Llama-3.1-8B LoRA r=8, bs=1, seq=%d, %d steps, CPU eager (oracle/ref.py, fp32)" % (a.seq, a.steps),
                      "kind": "port", "layers": a.layers, "cores": cores, "p50_step_s": round(p50, 3), "tokens_per_s": round(a.seq / p50, 2),
                      "first_step_s": round(times[0], 3), "init_s": round(init_s, 1), "losses": [round(x, 4) for x in losses]}))


if __name__ == "__main__":
    main()
