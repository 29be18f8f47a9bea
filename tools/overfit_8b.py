"""Sanity at full size: 8B-dimension model (random init), LoRA r=16, ONE fixed 4096-token batch, AdamW - the loss must fall.
(The tiny-model trajectory is pinned against the oracle in tests/test_model_gpu.py; this only shows the full-size path trains.)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
import torch
import bench
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model, cfg = bench.build_model("llama31_8b", 4096, 16, dev, "text")
from llx.arena import TrainableArena
from llx.train import Trainer
use_arena = os.environ.get("ARENA", "1") != "0"   # default: the flat trainable arena + llx.train.Trainer (as bench.py steps)
params = TrainableArena(model).params() if use_arena else [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=2e-3, weight_decay=0.0, fused=True)
trainer = Trainer(model, opt)
g = torch.Generator(device=dev); g.manual_seed(0)
ids = torch.randint(0, cfg.vocab_size, (1, 4096), device=dev, generator=g)
labels = torch.roll(ids, -1, 1); labels[:, -1] = -100
labels[:, :1024] = -100  # a masked prompt: the LM head and the loss run over the 3071 labelled rows only
losses = []
for step in range(int(os.environ.get("STEPS", "12"))):
    losses.append(float(trainer.step(lambda m: m(ids, labels=labels))))
    print(f"step {step:2d} loss {losses[-1]:.4f}", flush=True)
assert losses[-1] < losses[0] - 0.5, "the loss did not fall"
print("ok: loss fell from %.3f to %.3f (arena=%s)" % (losses[0], losses[-1], use_arena))
