"""Probe: do two half-chip GEMM launches (128 tiles each) on two HIP streams run concurrently?  And does a small kernel on a second
stream fill the idle half of the chip while a 128-tile GEMM runs?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
import torch  # noqa: E402

from llx import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
a = torch.randn(2048, 4096, device=dev).bfloat16()
b = torch.randn(4096, 4096, device=dev).bfloat16()
o1, o2 = torch.empty(2048, 4096, device=dev, dtype=torch.bfloat16), torch.empty(2048, 4096, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(4096, 4096, device=dev).bfloat16()
wa = torch.randn(16, 4096, device=dev).bfloat16()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def one():
    K.gemm_nt(a, b, out=o1)


def serial():
    K.gemm_nt(a, b, out=o1)
    K.gemm_nt(a, b, out=o2)


def parallel():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        K.gemm_nt(a, b, out=o1)
    with torch.cuda.stream(s2):
        K.gemm_nt(a, b, out=o2)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def gemm_then_skinny_serial():
    K.gemm_nt(a, b, out=o1)
    for _ in range(4):
        K.skinny_nt(x, wa)


def gemm_and_skinny_parallel():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        K.gemm_nt(a, b, out=o1)
    with torch.cuda.stream(s2):
        for _ in range(4):
            K.skinny_nt(x, wa)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


print(f"one 128-tile GEMM          {timed(one):8.1f} us")
print(f"two, same stream           {timed(serial):8.1f} us")
print(f"two, two streams           {timed(parallel):8.1f} us")
print(f"GEMM + 4 skinny, serial    {timed(gemm_then_skinny_serial):8.1f} us")
print(f"GEMM + 4 skinny, 2 streams {timed(gemm_and_skinny_parallel):8.1f} us")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gemm_and_skinny_parallel()
print(f"same, replayed from a hipGraph {timed(g.replay):8.1f} us")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    gemm_then_skinny_serial()
print(f"serial, replayed from a hipGraph {timed(g2.replay):8.1f} us")
