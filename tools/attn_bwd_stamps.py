"""In-kernel s_memtime stamps of the dK/dV kernel (workgroup 0 = heaviest key block, wave 0): cycles per segment of a query tile."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K, _lib as L
lib = L.load()
S, H, KVH = 4096, 32, 8
q = torch.randn(1, S, H, 128, device="cuda").bfloat16(); k = torch.randn(1, S, KVH, 128, device="cuda").bfloat16(); v = torch.randn(1, S, KVH, 128, device="cuda").bfloat16()
do = torch.randn_like(q); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
o, lse = K.attn_fwd(q, k, v)
for _ in range(3): K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv)
st = torch.zeros(1024, device="cuda", dtype=torch.int64)
L.check(lib.llx_debug_attn_bwd_set_stamps(L.ptr(st)), "set")
for _ in range(2): K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv)
torch.cuda.synchronize()
L.check(lib.llx_debug_attn_bwd_set_stamps(None), "unset")
t = st.cpu().tolist(); n = max(i for i, x in enumerate(t) if x) + 1
names = ["issue DMA", "S chain (8 MFMA)", "mask+softmax", "dP chain (8 MFMA)", "dS + pack", "phase 2 (16 MFMA)", "second 32 rows (32 MFMA)", "DMA wait", "barrier", "loop/next tile"]
seg = [[] for _ in names]
for i in range(3, n // 10 - 2):
    b = t[10 * i: 10 * i + 11]
    if len(b) < 11: break
    for j in range(10): seg[j].append(b[j + 1] - b[j])
tot = 0
for nm, v2 in zip(names, seg):
    m = statistics.median(v2); tot += m
    print(f"{nm:28s} median {m:7.0f}  min {min(v2):6d}  max {max(v2):6d}")
print(f"tile total (sum of medians) {tot:.0f} cycles; MFMA-pipe time of the tile = 64 x 32 = 2048")
