"""Probe: does a decode GEMV (nontemporal weight loads) run faster when its weight already sits in the 256 MiB Infinity Cache?
Per trial: a 1 GiB fill evicts everything, then (warm) a torch reduction reads the weight with default-policy loads, then the GEMV is timed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
import torch  # noqa: E402

from llx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
big = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
x = torch.randn(1, 4096, device=dev).bfloat16()
h = torch.randn(1, 14336, device=dev).bfloat16()
cases = {"wo 4096x4096 (33.5 MB)": (torch.randn(4096, 4096, device=dev).bfloat16(), x),
         "w2 4096x14336 (117 MB)": (torch.randn(4096, 14336, device=dev).bfloat16(), h),
         "w1 14336x4096 (117 MB)": (torch.randn(14336, 4096, device=dev).bfloat16(), x)}
for name, (w, xin) in cases.items():
    for mode in ("cold", "warm", "cold", "warm"):
        ts = []
        for _ in range(12):
            big.fill_(1)
            if mode == "warm":
                w.view(torch.int16).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.gemv([w], xin)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"{name:28s} {mode}: median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f} us", flush=True)
