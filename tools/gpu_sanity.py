"""First-contact check on the GPU box: does libllx_hip.so load into the torch process and run on torch's stream?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import ctypes
import torch
from llx import _lib as L

lib = L.load()
print("llx_version", lib.llx_version())
buf = ctypes.create_string_buffer(64)
print("CUs", lib.llx_device_info(0, buf, 64), buf.value)
torch.manual_seed(0)
for rows, dim in [(7, 512), (4096, 4096), (384, 512), (33, 1024)]:
    x = torch.randn(rows, dim, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(dim, device="cuda") * 0.1 + 1).bfloat16()
    y = torch.empty_like(x)
    rstd = torch.empty(rows, device="cuda", dtype=torch.float32)
    L.check(lib.llx_rmsnorm_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(rstd), rows, dim, 1e-5, L.stream()), "rmsnorm_fwd")
    xf = x.float()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w.float()).bfloat16()
    nbad = (y != ref).sum().item()
    print(f"rmsnorm_fwd {rows}x{dim}: mismatching elements {nbad} / {y.numel()}  maxdiff {(y.float()-ref.float()).abs().max().item():.3e}")
    dy = torch.randn_like(x)
    dx = torch.empty_like(x)
    dw = torch.zeros(dim, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(lib.llx_rmsnorm_bwd_workspace_bytes(rows, dim), device="cuda", dtype=torch.uint8)
    L.check(lib.llx_rmsnorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(w), L.ptr(rstd), L.ptr(dx), L.ptr(dw), 0, L.ptr(ws), None, rows, dim, L.stream()), "rmsnorm_bwd")
    xr = x.float().requires_grad_(); wr = w.float().requires_grad_()
    yr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * wr
    yr.backward(dy.float())
    print(f"   bwd dx maxdiff {(dx.float()-xr.grad).abs().max().item():.3e} (ref max {xr.grad.abs().max().item():.2f})  dw maxdiff {(dw.float()-wr.grad).abs().max().item():.3e} (ref max {wr.grad.abs().max().item():.2f})")
torch.cuda.synchronize()
# bandwidth
x = torch.randn(8192, 4096, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(x); w = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
rstd = torch.empty(8192, device="cuda")
for _ in range(3): lib.llx_rmsnorm_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(rstd), 8192, 4096, 1e-5, L.stream())
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): lib.llx_rmsnorm_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(rstd), 8192, 4096, 1e-5, L.stream())
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print(f"rmsnorm_fwd 8192x4096: {dt*1e6:.1f} us  {2*x.numel()*2/dt/1e12:.2f} TB/s")
print("SANITY DONE")
