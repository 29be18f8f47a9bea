import os, sys, math
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
import torch
from llx import kernels as K
from oracle import ref as O
from tests.test_kernels_gpu import _masks, _bf
cuda = torch.device("cuda")
for S in (640, 1024, 512):
    B, H, KVH, kind = 1, 8, 2, "docprefix"
    q = _bf(O.randn("q", (B, S, H, 128))); k = _bf(O.randn("k", (B, S, KVH, 128))); v = _bf(O.randn("v", (B, S, KVH, 128)))
    mask, doc, prefix = _masks(kind, B, S)
    ref = O.sdpa(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2), mask).transpose(1, 2)
    ms = K.MaskSpec(doc, prefix)
    o, lse = K.attn_fwd(q.to(cuda), k.to(cuda), v.to(cuda), ms)
    err = (o.cpu().float() - ref).abs().amax(dim=(0, 2, 3))
    bad = (err > 0.05).nonzero().view(-1)
    print("S", S, "prefix", prefix.tolist(), "bad rows", bad.numel(), bad[:20].tolist(), bad[-5:].tolist())
    nqb, nkt = (S + 127) // 128, (S + 63) // 64
    print(ms._flags.view(B, nqb, nkt).cpu())
    # reference classes
    m = mask[0, 0]
    cls = torch.zeros(nqb, nkt, dtype=torch.uint8)
    for i in range(nqb):
        for j in range(nkt):
            blk = m[i * 128 : (i + 1) * 128, j * 64 : (j + 1) * 64]
            cls[i, j] = 0 if not blk.any() else (2 if blk.all() else 1)
    print(cls)
