"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

The reference has no distributed code; the single exchange of a data-parallel step is the mean of the
TRAINABLE gradients (LoRA factors + norm weights: ~84 MB bf16 for Llama-3.1-8B r=16).  Gradients live in flat
contiguous bucket buffers (each ``param.grad`` is a view), buckets are filled in reverse layer order by backward and
each bucket's all-reduce is launched from a post-accumulate-grad hook as soon as its last gradient has landed, on
RCCL's own stream; ``finish()`` makes the optimizer stream wait.  xGMI is point-to-point: with payloads this small the
cost is latency, so few large buckets beat many small ones (default 4 buckets).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn


class GradBuckets:
    def __init__(self, model: nn.Module, n_buckets: int = 4, process_group=None, force: bool = False, overlap: bool = True,
                 groups: "list[list[nn.Parameter]] | None" = None):
        """force: build the flat buckets even for a single replica (tests / rehearsal of the N>1 path on one GPU).
        overlap: launch each bucket's all-reduce from backward hooks; False = exchange everything in finish() or bucket by bucket
        through launch() (used when forward+backward replay from captured hipGraphs, where hooks do not run).
        groups: explicit bucket contents in exchange order (StagedStep: one bucket per backward stage) instead of equal-size buckets."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or force
        # llx.arena.TrainableArena: the small trainables already live in one flat gradient buffer that backward writes in place -
        # their buckets are slices of it (no views to install, no accumulate-add kernels, no memset)
        self.arena = getattr(model, "_llx_arena", None)
        params = [p for p in model.parameters() if p.requires_grad]
        self._params = params
        if not self.active:
            # single replica: nothing to exchange - leave .grad to autograd (no flat views, no accumulate-add kernels)
            self.buckets, self._handles, self._hooks, self.sync_enabled, self.hook_launches = [], [], [], True, 0
            self.group_of_bucket = []
            return
        self.buckets: list[dict] = []
        self.group_of_bucket: list[int] = []
        if groups is not None:
            seen = {id(p) for g in groups for p in g}
            assert seen == {id(p) for p in params}, "groups must cover exactly the trainable parameters"
            for gi, g in enumerate(groups):
                by_kind: dict = {}
                for p in g:
                    by_kind.setdefault((p.dtype, self._in_arena(p)), []).append(p)
                for ps in by_kind.values():
                    self.buckets.append(self._make(ps))
                    self.group_of_bucket.append(gi)
        else:
            # backward produces gradients roughly in reverse registration order: bucket 0 = last layers
            params = list(reversed(params))
            total = sum(p.numel() for p in params)
            target = (total + n_buckets - 1) // max(1, n_buckets)
            cur, cur_n = [], 0
            for p in params:
                if cur and (p.dtype != cur[0].dtype or self._in_arena(p) != self._in_arena(cur[0])
                            or (cur_n + p.numel() > target and len(self.buckets) < n_buckets - 1)):
                    self.buckets.append(self._make(cur))
                    cur, cur_n = [], 0
                cur.append(p)
                cur_n += p.numel()
            if cur:
                self.buckets.append(self._make(cur))
        self._handles = []
        self._hooks = []
        self.hook_launches = 0  # all-reduces started from backward hooks (i.e. overlapped with the rest of backward)
        if overlap:
            for bi, b in enumerate(self.buckets):
                for p in b["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.sync_enabled = True  # set False on non-final micro-batches of gradient accumulation

    def _in_arena(self, p) -> bool:
        return self.arena is not None and self.arena.contains(p)

    def _make(self, params):
        dtype, device = params[0].dtype, params[0].device
        assert all(p.dtype == dtype for p in params), "a bucket holds one dtype"
        if self._in_arena(params[0]):
            flats = self.arena.ranges(params)  # normally ONE slice: the arena is laid out in layer order
            return {"params": params, "flat": flats[0], "flats": flats, "pending": len(params), "arena": True}
        flat = torch.zeros(sum(p.numel() for p in params), dtype=dtype, device=device)
        off = 0
        for p in params:
            p.grad = flat[off : off + p.numel()].view_as(p)
            off += p.numel()
        return {"params": params, "flat": flat, "flats": [flat], "pending": len(params), "arena": False}

    def _make_hook(self, bi: int):
        def hook(param):
            if not self.sync_enabled:
                return  # non-final micro-batch of an accumulation: gradients keep summing into the flat buffer, nothing is counted
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0 and not (b["flat"].is_cuda and torch.cuda.is_current_stream_capturing()):
                self.hook_launches += 1
                self._launch(b)

        return hook

    def prescale(self, b):
        """flat /= world (sum of means = mean of sums, keeps bf16 range).  Split from the exchange so that it can be captured at the
        end of a backward stage's hipGraph."""
        if not b.get("scaled"):
            if b["arena"]:
                self.arena.settle(b["params"])  # gradients that did not land in place (accumulation, foreign producers) join here
            if self.world > 1:
                for f in b["flats"]:
                    f.div_(self.world)
        b["scaled"] = True

    def _launch(self, b):
        b["launched"] = True
        self.prescale(b)
        if dist.is_initialized():
            for f in b["flats"]:
                self._handles.append(dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def launch_group(self, gi: int):
        """Start the exchange of every bucket of group `gi` now (asynchronous: RCCL's stream waits for the current stream's tail,
        the caller goes on queueing the next backward stage)."""
        if not (self.active and self.sync_enabled):
            return
        for b, g in zip(self.buckets, self.group_of_bucket):
            if g == gi and not b.get("launched"):
                self._launch(b)

    def finish(self):
        """Call after backward, before clip / optimizer.step: exchanges whatever the hooks have not launched yet
        (graph replay, parameters without a gradient this step) and waits for the in-flight bucket all-reduces."""
        if self.active and self.sync_enabled:
            for b in self.buckets:
                if not b.get("launched"):
                    self._launch(b)
        elif self.arena is not None and not self.active:
            self.arena.settle()  # single replica: the flat optimizer is the only reader of the arena's gradient buffer
        for h in self._handles:
            h.wait()
        self._handles.clear()
        for b in self.buckets:
            b["pending"] = len(b["params"])
            b["launched"] = False
            b["scaled"] = False

    def zero_grad(self):
        """Keep the views, zero the storage (optimizer.zero_grad(set_to_none=True) would drop the views)."""
        if not self.active:
            for p in self._params:
                p.grad = None
            return
        for b in self.buckets:
            if b["arena"]:
                for p in b["params"]:
                    p.grad = None  # the next backward writes the arena slice in place
            else:
                b["flat"].zero_()


class StagedStep:
    """Forward / backward cut into stages so that the gradient exchange of stage k runs under the backward of stage k-1 even when
    the kernels replay from hipGraphs (autograd hooks do not run in a replay, and an all-reduce issued after ONE whole-step graph
    overlaps with nothing).

        stages[0](*inputs) -> h0 ; stages[i](h_{i-1}) -> h_i ; stages[-1] returns the loss
        stage_params[i]: the trainable parameters whose gradients stage i's backward completes

    Activations are detached at the stage boundaries; backward runs last stage first, and as soon as stage i's backward is queued its
    bucket (one flat buffer per stage and dtype: GradBuckets(groups=...)) is pre-scaled and handed to RCCL, whose stream waits for
    that point only - the next stage's backward is queued right behind it on the compute stream.  With ``graph=True`` every piece
    (forward of all stages; backward of each stage incl. the pre-scale; optimizer step) is captured once into its own hipGraph
    sharing one memory pool, and a step is  replay(F), [replay(B_i), all_reduce_i]..., wait, replay(optimizer).
    Gradients are bit-identical to the un-staged step (same kernels, same order of accumulation)."""

    def __init__(self, model: nn.Module, stages, stage_params, optim, *, graph: bool = True, process_group=None, force: bool = False):
        self.stages, self.optim, self.use_graph = list(stages), optim, graph
        groups = [list(ps) for ps in reversed(list(stage_params))]  # exchange order = backward order
        self.buckets = GradBuckets(model, groups=groups, process_group=process_group, force=force, overlap=False)
        self.n = len(self.stages)
        self._graphs = None

    # ---- one eager pass: returns (loss, boundary tensors) - used for warm-up, for capture and as the eager step
    def _forward(self, inputs):
        self._ins, self._outs = [], []
        h = self.stages[0](*inputs)
        self._outs.append(h)
        for st in self.stages[1:]:
            hin = h.detach().requires_grad_(h.requires_grad)
            self._ins.append(hin)
            h = st(hin)
            self._outs.append(h)
        return h

    def _backward_stage(self, i: int):
        out = self._outs[i]
        if i == self.n - 1:
            out.backward()
        elif out.requires_grad:
            out.backward(self._ins[i].grad)
        if self.buckets.active:
            for b, g in zip(self.buckets.buckets, self.buckets.group_of_bucket):
                if g == self.n - 1 - i:
                    self.buckets.prescale(b)

    def _eager(self, inputs):
        self.buckets.zero_grad()
        loss = self._forward(inputs)
        for i in reversed(range(self.n)):
            self._backward_stage(i)
            self.buckets.launch_group(self.n - 1 - i)
        self.buckets.finish()
        self.optim.step()
        return loss.detach()

    def capture(self, *static_inputs):
        """Warm up (2 eager steps on a side stream), then capture the forward, every backward stage and the optimizer."""
        if not self.use_graph:
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._eager(static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        kw = dict(capture_error_mode="thread_local")  # the RCCL watchdog thread may poll events meanwhile
        gf = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf, **kw):
            self.buckets.zero_grad()  # captured memset of the flat buckets
            self._loss = self._forward(static_inputs)
        pool = gf.pool()
        gb = []
        for i in reversed(range(self.n)):
            for b in self.buckets.buckets:
                b["scaled"] = False
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, **kw):
                self._backward_stage(i)
            gb.append(g)
        go = torch.cuda.CUDAGraph()
        with torch.cuda.graph(go, pool=pool, **kw):
            self.optim.step()
        self._graphs = (gf, gb, go)

    def __call__(self, *inputs):
        """One optimizer step.  With graphs the inputs must already sit in the static tensors given to capture()."""
        if self._graphs is None:
            return self._eager(inputs)
        gf, gb, go = self._graphs
        gf.replay()
        for k, g in enumerate(gb):
            g.replay()
            for b, gi in zip(self.buckets.buckets, self.buckets.group_of_bucket):
                if gi == k:
                    b["scaled"] = True  # the replayed stage graph has pre-scaled this bucket
            self.buckets.launch_group(k)
        self.buckets.finish()
        go.replay()
        return self._loss


def llama_stages(model, n_stages: int, *, labels, block_mask=None, audio=None):
    """(stages, stage_params) for StagedStep over a modelling.Llama / LlamaAudio: contiguous runs of layers; the first stage also owns
    the embeddings (and the audio front end), the last one the final norm, the LM head and the loss."""
    L = len(model.layers)
    n_stages = max(1, min(n_stages, L))
    bounds = [round(i * L / n_stages) for i in range(n_stages + 1)]
    state = {}

    def first(tokens):
        x, n_drop = model._embed(tokens, audio) if audio is not None else model._embed(tokens)
        state["rope"], state["drop"] = model.rope[: x.shape[1]], n_drop
        x = model._run_layers(x, state["rope"], bounds[0], bounds[1], block_mask=block_mask)
        return head(x) if n_stages == 1 else x

    def head(x):
        if state["drop"]:
            x = x[:, state["drop"] :]
        return model._head(x, labels)

    def middle(lo, hi, last):
        def run(x):
            x = model._run_layers(x, state["rope"], lo, hi, block_mask=block_mask)
            return head(x) if last else x

        return run

    stages = [first] + [middle(bounds[i], bounds[i + 1], i == n_stages - 1) for i in range(1, n_stages)]
    layer_of = {id(p): li for li, layer in enumerate(model.layers) for p in layer.parameters()}
    stage_params = [[] for _ in range(n_stages)]
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if id(p) in layer_of:
            si = next(i for i in range(n_stages) if bounds[i] <= layer_of[id(p)] < bounds[i + 1])
        else:
            si = n_stages - 1 if name.startswith(("norm.", "output.")) else 0
        stage_params[si].append(p)
    return stages, stage_params
