"""w2 dgrad GEMM at the 8B shape: plain vs SwiGLU-backward epilogue (A/B across library builds with LLX_LIB_PATH)."""
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
M, I, D = 4096, 14336, 4096
dy = torch.randn(M, D, device="cuda").bfloat16(); wt = (torch.randn(I, D, device="cuda") * 0.02).bfloat16()
a2 = torch.randn(M, 64, device="cuda").bfloat16(); b2 = (torch.randn(I, 64, device="cuda") * 0.02).bfloat16()
gu = torch.randn(M, 2 * I, device="cuda").bfloat16(); dgu = torch.empty_like(gu); dh = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
name = os.path.basename(os.environ.get("LLX_LIB_PATH", "libllx_hip.so"))
print(name, f"plain {t(lambda: K.gemm_nt(dy, wt, out=dh, a2=a2, b2=b2)):.1f} us | swiglu-bwd epilogue {t(lambda: K.gemm_nt(dy, wt, out=dgu, a2=a2, b2=b2, epilogue=K.EPI_SWIGLU_BWD, e=gu)):.1f} us | stand-alone swiglu_bwd {t(lambda: K.swiglu_bwd(dh, gu[:, :I], gu[:, I:], dgu[:, :I], dgu[:, I:])):.1f} us | swiglu_fwd {t(lambda: K.swiglu_fwd(gu[:, :I], gu[:, I:])):.1f} us", flush=True)
