"""Probe: does an `external=True` event recorded INSIDE a captured hipGraph release a waiter on another stream while the rest of the
graph is still running?  (Mechanism for overlapping the RCCL gradient exchange with a replayed backward.)"""
import time

import torch

dev = torch.device("cuda:0")
a = torch.zeros(1 << 24, device=dev)
b = torch.ones(1 << 26, device=dev)
side = torch.cuda.Stream()
ev = torch.cuda.Event(external=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    a.add_(1)
    for _ in range(3):
        b.mul_(1.0001)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
a.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a.add_(1)
    ev.record()
    for _ in range(300):
        b.mul_(1.0001)
torch.cuda.synchronize()
for it in range(3):
    a.zero_()
    torch.cuda.synchronize()
    t_main_end = torch.cuda.Event(enable_timing=True)
    t_side_end = torch.cuda.Event(enable_timing=True)
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    g.replay()
    t_main_end.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        c = a.clone()
        t_side_end.record()
    torch.cuda.synchronize()
    print(f"iter {it}: a after the graph's first kernel = {c[0].item()} (want 1.0); side stream done at {t0.elapsed_time(t_side_end):.3f} ms, "
          f"graph done at {t0.elapsed_time(t_main_end):.3f} ms -> {'OVERLAP' if t0.elapsed_time(t_side_end) < 0.5 * t0.elapsed_time(t_main_end) else 'no overlap'}")
