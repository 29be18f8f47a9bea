"""GEMM epilogue cost at the 8B shapes: plain vs residual vs RoPE vs SwiGLU epilogues (A/B across builds with LLX_LIB_PATH)."""
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
name = os.path.basename(os.environ.get("LLX_LIB_PATH", "libllx_hip.so"))
M, D, I = 4096, 4096, 14336
x = torch.randn(M, D, device="cuda").bfloat16(); a2 = torch.randn(M, 64, device="cuda").bfloat16()
wq = (torch.randn(6144, D, device="cuda") * 0.02).bfloat16(); bq = (torch.randn(6144, 64, device="cuda") * 0.02).bfloat16()
table = torch.randn(4096, 64, 2, device="cuda"); oq = torch.empty(M, 6144, device="cuda", dtype=torch.bfloat16)
print(name, f"qkv fwd: plain {t(lambda: K.gemm_nt(x, wq, out=oq, a2=a2, b2=bq)):.1f} us | +rope {t(lambda: K.gemm_nt(x, wq, out=oq, a2=a2, b2=bq, rope=(table, 4096, 5120))):.1f} us", flush=True)
wo = (torch.randn(D, D, device="cuda") * 0.02).bfloat16(); bo = (torch.randn(D, 64, device="cuda") * 0.02).bfloat16(); res = torch.randn(M, D, device="cuda").bfloat16(); oo = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
print(name, f"wo fwd: plain {t(lambda: K.gemm_nt(x, wo, out=oo, a2=a2, b2=bo)):.1f} us | +residual {t(lambda: K.gemm_nt(x, wo, out=oo, a2=a2, b2=bo, epilogue=K.EPI_RESIDUAL, e=res)):.1f} us", flush=True)
w13 = (torch.randn(2 * I, D, device="cuda") * 0.02).bfloat16(); b13 = (torch.randn(2 * I, 64, device="cuda") * 0.02).bfloat16(); gu = torch.empty(M, 2 * I, device="cuda", dtype=torch.bfloat16); h = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
print(name, f"gate|up fwd: plain {t(lambda: K.gemm_nt(x, w13, out=gu, a2=a2, b2=b13)):.1f} us | +swiglu {t(lambda: K.gemm_nt(x, w13, out=gu, a2=a2, b2=b13, epilogue=K.EPI_SWIGLU_FWD, e=h)):.1f} us", flush=True)
w2t = (torch.randn(I, D, device="cuda") * 0.02).bfloat16(); b2t = (torch.randn(I, 64, device="cuda") * 0.02).bfloat16(); dgu = torch.empty_like(gu)
print(name, f"w2 dgrad: plain {t(lambda: K.gemm_nt(x, w2t, out=h, a2=a2, b2=b2t)):.1f} us | +swiglu bwd {t(lambda: K.gemm_nt(x, w2t, out=dgu, a2=a2, b2=b2t, epilogue=K.EPI_SWIGLU_BWD, e=gu)):.1f} us", flush=True)
