"""GEMM correctness + throughput on the GPU box (random data, interleaved runs)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import _lib as L
lib = L.load()

def gemm(a, b, c, a2=None, b2=None, epi=0, e=None):
    M, K = a.shape; N = b.shape[0]
    K2 = a2.shape[1] if a2 is not None else 0
    L.check(lib.llx_gemm_nt_bf16(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(c), c.stride(0), M, N, K,
        L.ptr(a2), a2.stride(0) if a2 is not None else 0, L.ptr(b2), b2.stride(0) if b2 is not None else 0, K2,
        epi, L.ptr(e), e.stride(0) if (e is not None and e.dim() == 2) else 0, L.stream()), "gemm")

torch.manual_seed(0)
dev = "cuda"
# ---- correctness
for (M, N, K, K2, epi) in [(256, 256, 64, 0, 0), (256, 256, 128, 0, 0), (512, 768, 256, 0, 0), (384, 1792, 512, 0, 0), (100, 520, 192, 64, 0),
                           (4096, 4096, 4096, 64, 1), (1000, 1024, 1792, 0, 2), (300, 264, 384, 0, 3), (4096, 1024, 4096, 0, 4)]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    a2 = torch.randn(M, K2, device=dev).bfloat16() if K2 else None
    b2 = (torch.randn(N, K2, device=dev) * 0.05).bfloat16() if K2 else None
    c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    e = None
    ref = a.float() @ b.float().T
    if K2: ref = ref + a2.float() @ b2.float().T
    if epi == 1:
        e = torch.randn(M, N, device=dev).bfloat16(); ref = ref.bfloat16().float() + e.float()
    elif epi == 2:
        e = torch.randn(N, device=dev).bfloat16(); ref = ref.bfloat16().float() + e.float()
    elif epi == 3:
        e = torch.randn(N, device=dev).bfloat16(); ref = torch.nn.functional.gelu((ref.bfloat16().float() + e.float()).bfloat16().float())
    elif epi == 4:
        e = (torch.rand(N, device=dev) * 0.1).bfloat16(); ref = ref.bfloat16().float() * e.float()
    gemm(a, b, c, a2, b2, epi, e)
    torch.cuda.synchronize()
    err = (c.float() - ref).abs().max().item(); scale = ref.abs().max().item()
    nbad = (c != ref.bfloat16()).sum().item()
    print(f"M={M} N={N} K={K} K2={K2} epi={epi}: maxerr {err:.4e} (ref max {scale:.2f}) nan={torch.isnan(c.float()).sum().item()} non-bitexact {nbad}/{c.numel()}", flush=True)

# ---- throughput (random data)
def bench(M, N, K, iters=20):
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): gemm(a, b, c)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): gemm(a, b, c)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    # torch (hipBLASLt) for context only
    for _ in range(3): torch.matmul(a, b.T)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): torch.matmul(a, b.T)
    e.record(); torch.cuda.synchronize()
    ms_t = s.elapsed_time(e) / iters
    fl = 2.0 * M * N * K
    print(f"gemm {M}x{N}x{K}: llx {ms*1e3:.1f} us {fl/ms/1e9:.0f} TF/s | hipblaslt {ms_t*1e3:.1f} us {fl/ms_t/1e9:.0f} TF/s", flush=True)

for shape in [(4096, 4096, 4096), (4096, 14336, 4096), (4096, 4096, 14336), (4096, 1024, 4096), (4096, 6144, 4096), (4096, 28672, 4096), (8192, 8192, 8192), (4096, 128256, 4096)]:
    bench(*shape)
print("GEMM BENCH DONE")
