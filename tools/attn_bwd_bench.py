"""Attention backward throughput on the GPU box (correctness lives in tests/test_kernels_gpu.py)."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
dev = "cuda"

def bench(B, S, H, KVH, iters=5):
    q = torch.randn(B, S, H, 128, device=dev).bfloat16(); k = torch.randn(B, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(B, S, KVH, 128, device=dev).bfloat16()
    do = torch.randn(B, S, H, 128, device=dev).bfloat16()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    o, lse = K.attn_fwd(q, k, v)
    for _ in range(2): K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv)
    e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e) / iters
    s.record()
    for _ in range(iters): K.attn_fwd(q, k, v)
    e.record(); torch.cuda.synchronize(); msf = s.elapsed_time(e) / iters
    fl = 4.0 * B * H * S * S * 128 / 2
    print(f"attn causal B={B} S={S} H={H}: fwd {msf*1e3:.1f} us {fl/msf/1e9:.0f} TF/s | bwd {ms*1e3:.1f} us {2.5*fl/ms/1e9:.0f} TF/s (algorithmic causal flops)", flush=True)
bench(1, 4096, 32, 8); bench(1, 8192, 32, 8)
