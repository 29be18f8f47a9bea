"""cProfile of the host side of one training step (8B, S=4096) to find Python/ctypes overhead."""
import cProfile, pstats, os, sys, io
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
import torch
import bench
dev = torch.device("cuda:0")
model, cfg = bench.build_model("llama31_8b", 4096, 16, dev)
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
ids = torch.randint(0, cfg.vocab_size, (1, 4096), device=dev); labels = torch.roll(ids, -1, 1)
def step():
    loss = model(ids, labels=labels); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize()
import time
t=time.perf_counter(); step(); t_host=time.perf_counter()-t; torch.cuda.synchronize(); t_all=time.perf_counter()-t
print(f"host-side time of one step {t_host*1e3:.1f} ms, until GPU done {t_all*1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
