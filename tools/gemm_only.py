import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
dev="cuda"
M,N,Kd=4096,4096,14336
a=torch.randn(M,Kd,device=dev).bfloat16(); b=torch.randn(N,Kd,device=dev).bfloat16(); c=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
for _ in range(4): K.gemm_nt(a,b,out=c)
torch.cuda.synchronize()
