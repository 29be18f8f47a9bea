import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
dev = "cuda"
B, S, H, KVH = 1, 4096, 32, 8
q = torch.randn(B, S, H, 128, device=dev).bfloat16(); k = torch.randn(B, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(B, S, KVH, 128, device=dev).bfloat16()
for _ in range(3): K.attn_fwd(q, k, v)
torch.cuda.synchronize()
