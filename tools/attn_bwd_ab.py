"""A/B timing of the attention backward across library builds (LLX_LIB_PATH), one process per build (run by the caller)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
S = int(os.environ.get("S", "4096")); H, KVH = 32, 8
q = torch.randn(1, S, H, 128, device="cuda").bfloat16(); k = torch.randn(1, S, KVH, 128, device="cuda").bfloat16(); v = torch.randn(1, S, KVH, 128, device="cuda").bfloat16()
do = torch.randn_like(q); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
o, lse = K.attn_fwd(q, k, v)
for _ in range(3): K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv)
torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv)
e.record(); torch.cuda.synchronize()
print(f"{os.path.basename(os.environ.get('LLX_LIB_PATH', 'libllx_hip.so')):28s} S={S} bwd {s.elapsed_time(e) / 20 * 1e3:.1f} us", flush=True)
