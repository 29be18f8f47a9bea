"""Skinny (LoRA) GEMM timing at the 8B shapes (A/B across library builds with LLX_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
name = os.path.basename(os.environ.get("LLX_LIB_PATH", "libllx_hip.so"))
for (M, Kd, R) in [(4096, 28672, 32), (4096, 14336, 16), (4096, 6144, 48), (4096, 4096, 64), (4096, 4096, 32), (4096, 4096, 16)]:
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = torch.randn(R, Kd, device="cuda").bfloat16()
    us = t(lambda: K.skinny_nt(x, w))
    print(f"{name} skinny_nt M={M} K={Kd} R={R}: {us:.1f} us  {M * Kd * 2 / us / 1e6:.2f} TB/s", flush=True)
for (M, N, R) in [(4096, 28672, 32), (4096, 4096, 64), (4096, 4096, 16), (4096, 14336, 16), (4096, 6144, 48)]:
    u = torch.randn(M, 64, device="cuda").bfloat16(); y = torch.randn(M, N, device="cuda").bfloat16(); out = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: K.skinny_tn(u, y, R, 1.0, out, False))
    print(f"{name} skinny_tn M={M} N={N} R={R}: {us:.1f} us  {M * N * 2 / us / 1e6:.2f} TB/s", flush=True)
for (M, Ns, r) in [(4096, (14336, 14336), 16), (4096, (4096, 1024, 1024), 16)]:
    Kd = sum(Ns); R = r * len(Ns)
    x = torch.randn(M, Kd, device="cuda").bfloat16(); w = torch.zeros(R, Kd, device="cuda", dtype=torch.bfloat16)
    kr = []; no = 0
    for i, n in enumerate(Ns):
        w[i * r:(i + 1) * r, no:no + n] = torch.randn(r, n, device="cuda").bfloat16(); kr += [no, no + n]; no += n
    kr += [0, 0] * (4 - len(Ns))
    print(f"{name} skinny_nt block-diagonal M={M} Ns={Ns}: dense {t(lambda: K.skinny_nt(x, w)):.1f} us, with k ranges {t(lambda: K.skinny_nt(x, w, kr)):.1f} us", flush=True)
