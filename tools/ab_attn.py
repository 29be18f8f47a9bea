"""Same-box A/B of the attention kernels: the current library against llama-x_amd/llx/libllx_hip_prev.so (tools/ab_build.sh <rev>),
launches interleaved (A B A B ...) so that both see the same clocks; per kernel the median of HIP-event times.
    python tools/ab_attn.py [fwd|bwd|both] [S ...]        LLX_AB_MASK=prefix|doc: a prefix-LM (P = S/2) / 17-document mask instead of causal"""
import ctypes, os, statistics, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
import torch
from llx import _lib as L
from llx import kernels as K

cur = L.load()
prev = ctypes.CDLL(os.path.join(ROOT, "llama-x_amd", "llx", "libllx_hip_prev.so"))
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(prev, name, None)
    if fn is not None:
        fn.restype, fn.argtypes = res, args
libs = {"prev": prev, "cur": cur}
what = sys.argv[1] if len(sys.argv) > 1 else "both"
sizes = [int(x) for x in sys.argv[2:]] or [4096]
dev = "cuda"
H, KVH = 32, 8
for S in sizes:
    q = torch.randn(1, S, H, 128, device=dev).bfloat16(); k = torch.randn(1, S, KVH, 128, device=dev).bfloat16(); v = torch.randn(1, S, KVH, 128, device=dev).bfloat16()
    do = torch.randn(1, S, H, 128, device=dev).bfloat16()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    mask = None
    mode = os.environ.get("LLX_AB_MASK", "")
    if mode == "prefix":
        mask = K.MaskSpec(prefix_len=torch.tensor([S // 2], device=dev, dtype=torch.int32))
    elif mode == "doc":
        bounds = torch.linspace(0, S, 18).long()
        ids = torch.zeros(S, dtype=torch.int32)
        for i in range(17): ids[bounds[i]:bounds[i + 1]] = i
        mask = K.MaskSpec(doc_ids=ids.to(dev).view(1, S))
    o, lse = K.attn_fwd(q, k, v, mask)
    times = {(n, w): [] for n in libs for w in ("fwd", "bwd")}
    outs = {}
    for it in range(24):
        for n, lib in libs.items():
            K._lib = lambda lib=lib: lib
            if what in ("fwd", "both"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); oo, ll = K.attn_fwd(q, k, v, mask); e1.record(); torch.cuda.synchronize()
                times[(n, "fwd")].append(e0.elapsed_time(e1) * 1e3); outs[(n, "fwd")] = oo
            if what in ("bwd", "both"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); K.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, mask); e1.record(); torch.cuda.synchronize()
                times[(n, "bwd")].append(e0.elapsed_time(e1) * 1e3); outs[(n, "bwd")] = (dq.clone(), dk.clone(), dv.clone())
    for w in ("fwd", "bwd"):
        if times[("cur", w)]:
            a, b = statistics.median(times[("prev", w)][4:]), statistics.median(times[("cur", w)][4:])
            same = (torch.equal(outs[("prev", w)], outs[("cur", w)]) if w == "fwd" else all(torch.equal(x, y) for x, y in zip(outs[("prev", w)], outs[("cur", w)])))
            print(f"S={S} {w}: prev {a:7.1f} us   cur {b:7.1f} us   ({(b / a - 1) * 100:+.1f} %)   outputs bit-identical: {same}", flush=True)
