"""Same-box A/B of the GEMM kernels on the training step's shapes: current library against libllx_hip_prev.so (tools/ab_build.sh
<rev>), launches alternating; median of HIP-event times.   python tools/ab_gemm.py [bf16|i8|all]"""
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
what = sys.argv[1] if len(sys.argv) > 1 else "bf16"
def bench(name, fn, flops):
    times = {n: [] for n in libs}; outs = {}
    for it in range(16):
        for n, lib in libs.items():
            K._lib = lambda lib=lib: lib
            L.load = lambda lib=lib: lib  # subclasses.int8_mm goes through L.load()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); o = fn(); e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) * 1e3); outs[n] = o
    a, b = statistics.median(times["prev"][4:]), statistics.median(times["cur"][4:])
    print(f"{name:60s} prev {a:7.1f} us  cur {b:7.1f} us  ({(b / a - 1) * 100:+.1f} %)  {flops / b / 1e6:6.0f} TF/s  bit-identical: {torch.equal(outs['prev'], outs['cur'])}", flush=True)

g = torch.Generator(device=dev); g.manual_seed(0)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
M, D, I = 4096, 4096, 14336
if what in ("bf16", "all"):
    x = rn(M, D); wo = rn(D, D); res = rn(M, D)
    bench("o proj + residual (EPI 1) 4096x4096x4096", lambda: K.gemm_nt(x, wo, epilogue=K.EPI_RESIDUAL, e=res), 2.0 * M * D * D)
    h = rn(M, I); w2 = rn(D, I)
    bench("w2 + residual (EPI 1) 4096x4096x14336", lambda: K.gemm_nt(h, w2, epilogue=K.EPI_RESIDUAL, e=res), 2.0 * M * D * I)
    dy = rn(M, D); w2t = rn(I, D); gu = rn(M, 2 * I); dgu = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16)
    bench("w2 dgrad + SwiGLU backward (EPI 6) 4096x14336x4096", lambda: K.gemm_nt(dy, w2t, out=dgu, epilogue=K.EPI_SWIGLU_BWD, e=gu), 2.0 * M * D * I)
    wqkv = rn(6144, D)
    bench("plain 4096x6144x4096", lambda: K.gemm_nt(x, wqkv), 2.0 * M * 6144 * D)
    w13 = rn(2 * I, D); hh = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    bench("gate|up + SwiGLU forward (EPI 7) 4096x28672x4096", lambda: K.gemm_nt(x, w13, epilogue=K.EPI_SWIGLU_FWD, e=hh), 2.0 * M * 2 * I * D)
if what in ("i8", "all"):
    from subclasses.int8_mm import _launch as i8_gemm
    ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int8)
    xi = ri(M, D); xs = (torch.rand(M, device=dev, generator=g) * 0.02).bfloat16()
    for name, N, epi in (("int8_mm_dequant 4096x4096x4096", D, 0), ("int8 gate|up + LoRA ext + SwiGLU (EPI 7) 4096x28672x4096", 2 * I, 7),
                         ("int8 q|k|v + LoRA ext 4096x6144x4096", 6144, 0), ("int8 wo + LoRA ext + residual (EPI 1)", D, 1)):
        w = ri(N, D); ws = (torch.rand(N, device=dev, generator=g) * 0.02).bfloat16()
        a2 = rn(M, 64) if "LoRA" in name else None
        b2 = rn(N, 64) if "LoRA" in name else None
        e = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16) if epi == 7 else (rn(M, N) if epi == 1 else None)
        bench(name, lambda: i8_gemm(xi, w, xs, ws, a2=a2, b2=b2, epilogue=epi, e=e), 2.0 * M * N * D)
