"""Same-box A/B of the skinny (adapter) kernels on the training step's shapes: current library against libllx_hip_prev.so
(tools/ab_build.sh <rev>), launches alternating; median of HIP-event times."""
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
dev = "cuda"
M = 4096
def bench(name, fn):
    times = {n: [] for n in libs}; outs = {}
    for it in range(30):
        for n, lib in libs.items():
            K._lib = lambda lib=lib: lib
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) * 1e3); outs[n] = o
    a, b = statistics.median(times["prev"][5:]), statistics.median(times["cur"][5:])
    same = all(torch.equal(x, y) for x, y in zip(outs["prev"], outs["cur"])) if isinstance(outs["cur"], (tuple, list)) else torch.equal(outs["prev"], outs["cur"])
    print(f"{name:58s} prev {a:7.1f} us  cur {b:7.1f} us  ({(b / a - 1) * 100:+.1f} %)  bit-identical: {same}", flush=True)

g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
# u = dy @ B of the fused groups (block-diagonal B^T, ranged): q|k|v (N=6144, R=48), gate|up (N=28672, R=32)
for N, ranks, ns in ((6144, (16, 16, 16), (4096, 1024, 1024)), (28672, (16, 16), (14336, 14336))):
    R = sum(ranks); bT = torch.zeros(64, N, device=dev, dtype=torch.bfloat16); kr = []
    ro = no = 0
    for r, n in zip(ranks, ns):
        bT[ro:ro + r, no:no + n] = rn(r, n); ro += r; no += n
    for nb in range(4):
        lo = hi = None; ro = no = 0
        for r, n in zip(ranks, ns):
            if ro < 16 * nb + 16 and ro + r > 16 * nb:
                lo = no if lo is None else min(lo, no); hi = no + n if hi is None else max(hi, no + n)
            ro += r; no += n
        kr += [lo or 0, hi or 0] if lo is not None else [0, 0]
    dy = rn(M, N)
    bench(f"skinny_nt dy[{M},{N}] . B^T (R={R}, ranged)", lambda: K.skinny_nt(dy, bT[:R], kr))
for Kd in (4096, 14336):
    x = rn(M, Kd); a = rn(16, Kd)
    bench(f"skinny_nt x[{M},{Kd}] . A^T (R=16)", lambda: K.skinny_nt(x, a))
x = rn(M, 4096); a48 = rn(48, 4096)
bench(f"skinny_nt x[{M},4096] . A^T (R=48, dense)", lambda: K.skinny_nt(x, a48))
x = rn(M, 4096); w = (1 + 0.1 * torch.randn(4096, device=dev, generator=g)).bfloat16()
for R in (48, 32):
    a = rn(R, 4096)
    bench(f"rmsnorm_skinny_nt x[{M},4096], R={R}", lambda: K.rmsnorm_skinny_nt(x, w, 1e-5, a))
# weight-gradient side: out[R, N] = scale * u^T . y (first stage: split partials; second stage: the reduce over the splits)
for N, R in ((4096, 16), (4096, 48), (14336, 16), (6144, 48), (28672, 32)):
    u = rn(M, 64); y = rn(M, N); out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
    bench(f"skinny_tn u[{M},64]^T . y[{M},{N}] (R={R}, both stages)", lambda: K.skinny_tn(u, y, R, 1.0, out, False))
