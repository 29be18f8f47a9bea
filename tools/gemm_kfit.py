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
for N in (4096, 28672):
    for Kd in (128, 512, 1024, 2048, 4096, 8192):
        a = torch.randn(4096, Kd, device="cuda").bfloat16(); b = (torch.randn(N, Kd, device="cuda") * 0.05).bfloat16(); out = torch.empty(4096, N, device="cuda", dtype=torch.bfloat16)
        us = t(lambda: K.gemm_nt(a, b, out=out))
        # graph replay timing (no host launch gaps)
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(2): K.gemm_nt(a, b, out=out)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=st):
                for _ in range(10): K.gemm_nt(a, b, out=out)
        ug = t(lambda: g.replay(), 5) / 10
        print(f"N={N} K={Kd:5d} ({Kd // 64:3d} K-tiles): eager {us:7.1f} us, in graph {ug:7.1f} us", flush=True)
