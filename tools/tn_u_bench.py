"""First stage of the adapter-gradient product dB = s t^T.dy with and without the fused u = dy @ B partials, on the gate|up shape."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llama-x_amd"))
import torch
from llx import kernels as K
dev = "cuda"; M = 4096
g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
for N, ranks, ns in ((28672, (16, 16), (14336, 14336)), (6144, (16, 16, 16), (4096, 1024, 1024)), (4096, (16,), (4096,))):
    R = sum(ranks)
    bT = torch.zeros(R, N, device=dev, dtype=torch.bfloat16); segs = []; ro = no = 0
    for r, n in zip(ranks, ns):
        bT[ro:ro + r, no:no + n] = rn(r, n); segs.append((no, no + n, ro, ro + r)); ro += r; no += n
    dy = rn(M, N); t = torch.zeros(M, 64, device=dev, dtype=torch.bfloat16); t[:, :R] = rn(M, R)
    flat = torch.empty(sum((b - a) * (d - c) for a, b, c, d in segs), device=dev, dtype=torch.bfloat16)
    def run(fused):
        pend = []
        K.skinny_tn(t, dy, R, 1.0, flat, transpose_out=True, segs=segs, pending=pend, u_from=bT if fused else None)
        u = K.skinny_u_reduce(pend[-1]) if fused else None
        K.skinny_tn_flush(pend)
        return u
    for fused in (False, True):
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); u = run(fused); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"N={N} R={R} fused={fused}: {statistics.median(ts[5:]):7.1f} us (first stage + reduces)", flush=True)
    kr = []
    for nb in range(4):
        hit = [(a, b) for (a, b, c, d) in segs if c < 16 * nb + 16 and d > 16 * nb]
        kr += [min(h[0] for h in hit), max(h[1] for h in hit)] if hit else [0, 0]
    ref = K.skinny_nt(dy, bT, kr if len(segs) > 1 else None)
    u = run(True)
    print("   max |u_fused - u_skinny_nt| =", (u.float() - ref.float()).abs().max().item(), " max|u| =", ref.float().abs().max().item())
