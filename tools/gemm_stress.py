"""Race / edge screening of the GEMM kernel: many shapes (ragged M/N, K-extension, epilogues), each run several times and compared
bit-for-bit across runs and against fp32 matmul."""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
dev = "cuda"; random.seed(0); torch.manual_seed(0)
bad = 0
shapes = [(4096, 4096, 4096, 0), (4096, 6144, 4096, 64), (4096, 28672, 4096, 64), (4096, 4096, 28672, 64), (4096, 4096, 64, 0), (4096, 4096, 128, 64), (256, 256, 192, 0)]
for _ in range(40):
    shapes.append((random.choice([1, 37, 255, 256, 257, 1000, 2048, 3000]), 8 * random.randint(1, 300), 64 * random.randint(1, 40), random.choice([0, 0, 64, 128])))
for (M, N, Kd, K2) in shapes:
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = (torch.randn(N, Kd, device=dev) * 0.05).bfloat16()
    a2 = torch.randn(M, K2, device=dev).bfloat16() if K2 else None
    b2 = (torch.randn(N, K2, device=dev) * 0.05).bfloat16() if K2 else None
    e = torch.randn(M, N, device=dev).bfloat16()
    ref = a.float() @ b.float().T
    if K2: ref = ref + a2.float() @ b2.float().T
    ref = ref.bfloat16().float() + e.float()
    outs = [K.gemm_nt(a, b, a2=a2, b2=b2, epilogue=K.EPI_RESIDUAL, e=e) for _ in range(4)]
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    err = (outs[0].float() - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
    ok = same and err < 2 ** -6 and not torch.isnan(outs[0].float()).any().item()
    if not ok:
        bad += 1
        print("FAIL", M, N, Kd, K2, "same", same, "relerr", err)
print("gemm stress:", len(shapes), "shapes,", bad, "failures")
sys.exit(1 if bad else 0)
