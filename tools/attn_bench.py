"""Attention forward correctness (vs fp32 math) + throughput on the GPU box."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import _lib as L
lib = L.load()
dev = "cuda"

def attn_fwd(q, k, v, doc=None, prefix=None):
    B, S, H, hd = q.shape; KVH = k.shape[2]
    o = torch.empty(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    flags = None
    if doc is not None or prefix is not None:
        flags = torch.empty(lib.llx_attn_flags_bytes(B, S), device=dev, dtype=torch.uint8)
        L.check(lib.llx_attn_tile_flags(L.ptr(doc), L.ptr(prefix), L.ptr(flags), B, S, L.stream()), "flags")
    L.check(lib.llx_attn_fwd(L.ptr(q), q.stride(0), q.stride(1), L.ptr(k), k.stride(0), k.stride(1), L.ptr(v), v.stride(0), v.stride(1),
                             L.ptr(o), o.stride(0), o.stride(1), L.ptr(lse), L.ptr(doc), L.ptr(prefix), L.ptr(flags), B, S, H, KVH, hd,
                             1.0 / math.sqrt(hd), L.stream()), "attn_fwd")
    return o, lse

def ref(q, k, v, mask):
    B, S, H, hd = q.shape; g = H // k.shape[2]
    qf = q.float().transpose(1, 2); kf = k.float().transpose(1, 2).repeat_interleave(g, 1); vf = v.float().transpose(1, 2).repeat_interleave(g, 1)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
    s = s.masked_fill(~mask, float("-inf"))
    lse = torch.logsumexp(s, -1)
    return (torch.softmax(s, -1) @ vf).transpose(1, 2), lse

torch.manual_seed(0)
for (B, S, H, KVH, kind) in [(1, 256, 4, 1, "causal"), (2, 384, 4, 1, "causal"), (1, 200, 8, 2, "causal"), (1, 512, 4, 1, "doc"), (2, 384, 4, 2, "prefix"), (1, 1024, 8, 2, "docprefix")]:
    q = torch.randn(B, S, H, 128, device=dev).bfloat16(); k = torch.randn(B, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(B, S, KVH, 128, device=dev).bfloat16()
    idx = torch.arange(S, device=dev)
    mask = (idx[:, None] >= idx[None, :])[None, None].expand(B, 1, S, S).clone()
    doc = prefix = None
    if "prefix" in kind:
        prefix = torch.tensor([S // 3, S // 2][:B] + [7] * (B - 2), device=dev, dtype=torch.int32)[:B]
        mask = mask | (idx[None, None, None, :] < prefix.view(B, 1, 1, 1))
    if "doc" in kind:
        cuts = sorted(torch.randint(1, S - 1, (5,)).tolist())
        d = torch.zeros(S, dtype=torch.int32)
        for c in cuts: d[c:] += 1
        d[S - 37:] = 0   # the reference's tail quirk: unused tail carries id 0
        doc = d.to(dev).view(1, S).expand(B, S).contiguous()
        mask = mask & (doc[:, None, :, None] == doc[:, None, None, :])
    o, lse = attn_fwd(q, k, v, doc, prefix)
    ro, rl = ref(q, k, v, mask)
    torch.cuda.synchronize()
    print(f"B={B} S={S} H={H} KVH={KVH} {kind}: o maxerr {(o.float()-ro).abs().max().item():.3e} (ref max {ro.abs().max().item():.2f}) lse maxerr {(lse*math.log(2)-rl).abs().max().item():.3e} nan {torch.isnan(o.float()).sum().item()}", flush=True)

# strided views from a fused qkv buffer
B, S, H, KVH = 1, 512, 8, 2
qkv = torch.randn(B, S, (H + 2 * KVH) * 128, device=dev).bfloat16()
q = qkv[..., : H * 128].view(B, S, H, 128); k = qkv[..., H * 128 : (H + KVH) * 128].view(B, S, KVH, 128); v = qkv[..., (H + KVH) * 128 :].view(B, S, KVH, 128)
idx = torch.arange(S, device=dev); mask = (idx[:, None] >= idx[None, :])[None, None]
o, lse = attn_fwd(q, k, v); ro, rl = ref(q, k, v, mask)
print(f"strided qkv views: o maxerr {(o.float()-ro).abs().max().item():.3e}")

def bench(B, S, H, KVH, iters=10):
    q = torch.randn(B, S, H, 128, device=dev).bfloat16(); k = torch.randn(B, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(B, S, KVH, 128, device=dev).bfloat16()
    for _ in range(2): attn_fwd(q, k, v)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): attn_fwd(q, k, v)
    e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e) / iters
    fl = 4.0 * B * H * S * S * 128 / 2
    print(f"attn fwd causal B={B} S={S} H={H}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s (causal flops)", flush=True)
bench(1, 4096, 32, 8); bench(1, 8192, 32, 8); bench(4, 2048, 32, 8)
print("ATTN DONE")
