"""TN (weight-gradient) GEMM timing against the transposed-copies + NT route it replaces."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for (M, N1, N2) in [(4096, 4096, 4096), (4096, 14336, 4096), (4096, 128256, 4096), (2048, 4096, 12288)]:
    a = torch.randn(M, N1, device="cuda").bfloat16(); b = torch.randn(M, N2, device="cuda").bfloat16()
    tn = t(lambda: K.gemm_tn(a, b))
    old = t(lambda: K.gemm_nt(K.transpose(a, 64), K.transpose(b, 64)))
    fl = 2.0 * M * N1 * N2
    print(f"M={M} N1={N1} N2={N2}: TN kernel {tn:.1f} us ({fl / tn / 1e6:.0f} TF/s) | transposes + NT {old:.1f} us ({fl / old / 1e6:.0f} TF/s)", flush=True)
