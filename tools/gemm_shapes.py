"""bf16 GEMM timing at the 8B step shapes (A/B knobs through the environment, e.g. LLX_GEMM_PERSISTENT=0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("LLX_GEMM")) or "default"
tot = 0.0
for (name, M, N, Kd, K2, per_step) in [("qkv fwd", 4096, 6144, 4096, 64, 32), ("wo fwd/dgrad", 4096, 4096, 4096, 64, 64), ("gate|up fwd", 4096, 28672, 4096, 64, 32), ("w2 fwd", 4096, 4096, 14336, 64, 32),
                                        ("w2 dgrad", 4096, 14336, 4096, 64, 32), ("gate|up dgrad", 4096, 4096, 28672, 64, 32), ("qkv dgrad", 4096, 4096, 6144, 64, 32), ("head fwd", 4096, 128256, 4096, 0, 1), ("head dgrad", 4096, 4096, 128256, 0, 1)]:
    a = torch.randn(M, Kd, device="cuda").bfloat16(); b = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16()
    a2 = torch.randn(M, K2, device="cuda").bfloat16() if K2 else None; b2 = (torch.randn(N, K2, device="cuda") * 0.05).bfloat16() if K2 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: K.gemm_nt(a, b, out=out, a2=a2, b2=b2), 10 if N > 100000 or Kd > 100000 else 20)
    tot += us * per_step
    print(f"[{tag}] {name:14s} tiles={((M + 255) // 256) * ((N + 255) // 256):5d} K={Kd + K2:6d}: {us:8.1f} us ({2.0 * M * N * (Kd + K2) / us / 1e6:5.0f} TF/s)", flush=True)
    del a, b, out
print(f"[{tag}] sum over a step's launches: {tot / 1e3:.2f} ms", flush=True)
