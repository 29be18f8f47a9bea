import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import _lib as L
lib = L.load(); dev = "cuda"
S, H, KVH = 4096, 32, 8
q = torch.randn(1, S, H, 128, device=dev).bfloat16(); k = torch.randn(1, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(1, S, KVH, 128, device=dev).bfloat16()
o = torch.empty_like(q); st = torch.zeros(512, device=dev, dtype=torch.int64)
for _ in range(2):
    L.check(lib.llx_debug_attn_fwd_stamps(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), S, H, KVH, L.ptr(st), L.stream()), "stamps")
torch.cuda.synchronize()
t = st.cpu().tolist()
n = max(i for i, x in enumerate(t) if x) + 1
t = t[:n]
print("stamps", n, "tiles", n // 5)
import statistics
seg = {k: [] for k in ("qk", "softmax", "pv", "loadwait", "barrier+next")}
for i in range(2, n // 5 - 2):  # skip the first tiles
    b = t[5 * i: 5 * i + 6]
    if len(b) < 6: break
    seg["qk"].append(b[1] - b[0]); seg["softmax"].append(b[2] - b[1]); seg["pv"].append(b[3] - b[2]); seg["loadwait"].append(b[4] - b[3]); seg["barrier+next"].append(b[5] - b[4])
for k2, v2 in seg.items():
    print(f"{k2:14s} median {statistics.median(v2):8.0f} cycles  min {min(v2):6d} max {max(v2):6d}")
print("tile total median", statistics.median([t[5 * i + 5] - t[5 * i] for i in range(2, n // 5 - 2)]))
