import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
dev="cuda"
def bench(M,N,Kd,iters=20):
    a=torch.randn(M,Kd,device=dev).bfloat16(); b=torch.randn(N,Kd,device=dev).bfloat16(); c=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    for _ in range(3): K.gemm_nt(a,b,out=c)
    ref=(a[:64].float()@b.float().T)
    err=(c[:64].float()-ref).abs().max().item()/ref.abs().max().item()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): K.gemm_nt(a,b,out=c)
    e.record(); torch.cuda.synchronize(); ms=s.elapsed_time(e)/iters
    print(f"gemm {M}x{N}x{Kd}: {ms*1e3:.1f} us {2.0*M*N*Kd/ms/1e9:.0f} TF/s relerr {err:.1e}", flush=True)
for sh in [(4096,4096,4096),(4096,6144,4096),(4096,28672,4096),(4096,4096,28672),(4096,14336,4096),(4096,4096,14336),(8192,8192,8192)]: bench(*sh)
