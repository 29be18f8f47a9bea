"""Fused RMSNorm + adapter projection (llx_rmsnorm_skinny_nt) timing at the 8B shape (A/B across library builds with LLX_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
T, D = 4096, 4096
x = torch.randn(T, D, device="cuda").bfloat16(); w = torch.randn(D, device="cuda").bfloat16()
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
name = os.path.basename(os.environ.get("LLX_LIB_PATH", "libllx_hip.so"))
for R in (16, 32, 48):
    a = torch.randn(R, D, device="cuda").bfloat16()
    print(name, f"R={R}: norm+skinny {t(lambda: K.rmsnorm_skinny_nt(x, w, 1e-5, a)):.1f} us | norm {t(lambda: K.rmsnorm_fwd(x, w, 1e-5)):.1f} + skinny {t(lambda: K.skinny_nt(x, a)):.1f} us", flush=True)
