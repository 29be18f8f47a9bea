"""Attention backward correctness (vs fp32 autograd) + throughput on the GPU box."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import _lib as L
lib = L.load()
dev = "cuda"

def mkflags(doc, prefix, B, S):
    if doc is None and prefix is None: return None
    flags = torch.empty(lib.llx_attn_flags_bytes(B, S), device=dev, dtype=torch.uint8)
    L.check(lib.llx_attn_tile_flags(L.ptr(doc), L.ptr(prefix), L.ptr(flags), B, S, L.stream()), "flags")
    return flags

def attn_fwd(q, k, v, doc=None, prefix=None, flags=None):
    B, S, H, hd = q.shape; KVH = k.shape[2]
    o = torch.empty(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    L.check(lib.llx_attn_fwd(L.ptr(q), q.stride(0), q.stride(1), L.ptr(k), k.stride(0), k.stride(1), L.ptr(v), v.stride(0), v.stride(1),
                             L.ptr(o), o.stride(0), o.stride(1), L.ptr(lse), L.ptr(doc), L.ptr(prefix), L.ptr(flags), B, S, H, KVH, hd,
                             1.0 / math.sqrt(hd), L.stream()), "attn_fwd")
    return o, lse

def attn_bwd(q, k, v, o, do, lse, doc=None, prefix=None, flags=None):
    B, S, H, hd = q.shape; KVH = k.shape[2]
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    delta = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    L.check(lib.llx_attn_bwd(L.ptr(q), q.stride(0), q.stride(1), L.ptr(k), k.stride(0), k.stride(1), L.ptr(v), v.stride(0), v.stride(1),
        L.ptr(o), o.stride(0), o.stride(1), L.ptr(do), do.stride(0), do.stride(1), L.ptr(lse), L.ptr(delta),
        L.ptr(dq), dq.stride(0), dq.stride(1), L.ptr(dk), dk.stride(0), dk.stride(1), L.ptr(dv), dv.stride(0), dv.stride(1),
        L.ptr(doc), L.ptr(prefix), L.ptr(flags), B, S, H, KVH, hd, 1.0 / math.sqrt(hd), L.stream()), "attn_bwd")
    return dq, dk, dv

def ref(q, k, v, mask, do):
    B, S, H, hd = q.shape; g = H // k.shape[2]
    q = q.float().requires_grad_(); k = k.float().requires_grad_(); v = v.float().requires_grad_()
    qf = q.transpose(1, 2); kf = k.transpose(1, 2).repeat_interleave(g, 1); vf = v.transpose(1, 2).repeat_interleave(g, 1)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
    s = s.masked_fill(~mask, float("-inf"))
    o = (torch.softmax(s, -1) @ vf).transpose(1, 2)
    o.backward(do.float())
    return q.grad, k.grad, v.grad

torch.manual_seed(0)
for (B, S, H, KVH, kind) in [(1, 256, 4, 1, "causal"), (2, 384, 4, 1, "causal"), (1, 200, 8, 2, "causal"), (1, 512, 4, 1, "doc"), (2, 384, 4, 2, "prefix"), (1, 1024, 8, 2, "docprefix")]:
    q = torch.randn(B, S, H, 128, device=dev).bfloat16(); k = torch.randn(B, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(B, S, KVH, 128, device=dev).bfloat16()
    do = torch.randn(B, S, H, 128, device=dev).bfloat16()
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
        d[S - 37:] = 0
        doc = d.to(dev).view(1, S).expand(B, S).contiguous()
        mask = mask & (doc[:, None, :, None] == doc[:, None, None, :])
    flags = mkflags(doc, prefix, B, S)
    o, lse = attn_fwd(q, k, v, doc, prefix, flags)
    dq, dk, dv = attn_bwd(q, k, v, o, do, lse, doc, prefix, flags)
    rq, rk, rv = ref(q, k, v, mask, do)
    torch.cuda.synchronize()
    f = lambda x, y: f"{(x.float()-y).abs().max().item():.3e}/{y.abs().max().item():.2f}"
    print(f"B={B} S={S} H={H} KVH={KVH} {kind}: dq {f(dq, rq)}  dk {f(dk, rk)}  dv {f(dv, rv)}  nan {sum(torch.isnan(t.float()).sum().item() for t in (dq, dk, dv))}", flush=True)

def bench(B, S, H, KVH, iters=5):
    q = torch.randn(B, S, H, 128, device=dev).bfloat16(); k = torch.randn(B, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(B, S, KVH, 128, device=dev).bfloat16()
    do = torch.randn(B, S, H, 128, device=dev).bfloat16()
    o, lse = attn_fwd(q, k, v)
    for _ in range(2): attn_bwd(q, k, v, o, do, lse)
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): attn_bwd(q, k, v, o, do, lse)
    e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e) / iters
    fl = 2.5 * 4.0 * B * H * S * S * 128 / 2
    print(f"attn bwd causal B={B} S={S} H={H}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s (algorithmic 5-product causal flops)", flush=True)
bench(1, 4096, 32, 8); bench(1, 8192, 32, 8)
print("ATTN BWD DONE")
