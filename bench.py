"""Headline benchmark: train tokens/sec of Llama-3.1-8B, LoRA r=16, bf16, seq 4096, on N MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + backward + (gradient all-reduce) + AdamW step over one synthetic 4096-token batch per GPU
(config C2 of SURVEY 8d: text only, causal, base + LM head + embeddings frozen, LoRA on model.layers, norm weights
trainable).  Weights are random-init at Llama-3.1-8B dimensions (no checkpoints offline).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: RCCL's device-memory sharing across the ranks of a node needs it (the launcher's
# environment normally carries it already; a bare shell may not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic FLOPs per token of the LoRA training step at S=4096: fwd + dgrad through every base linear
# and the frozen LM head, causal attention at half, recompute not counted
GF_PER_TOKEN = {4096: 34.0, 8192: 37.8}
BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def build_model(cfg_name: str, seq: int, rank: int, device, config: str = "text", trainable: str = "lora"):
    from modelling import Llama, LlamaAudio, LlamaConfig, apply_linear_adapter_
    from subclasses import quantize_linear_

    if cfg_name == "llama31_8b":
        cfg = LlamaConfig(embed_dim=4096, num_layers=32, head_dim=128, num_heads=32, num_kv_heads=8, intermediate_dim=14336,
                          max_seq_len=seq, vocab_size=128_256, rope_base=500_000, is_llama3_1=True)
    else:  # small config for plumbing checks
        cfg = LlamaConfig(embed_dim=512, num_layers=2, head_dim=128, num_heads=4, num_kv_heads=1, intermediate_dim=1792,
                          max_seq_len=seq, vocab_size=1024, rope_base=500_000, is_llama3_1=True)
    with torch.device("meta"):
        model = (LlamaAudio if config == "audio" else Llama)(cfg)
    model = model.to(torch.bfloat16).to_empty(device=device)
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)
    model.build_cache()
    model.rope = model.rope.to(device)
    if config == "audio":
        model.melspec = model.melspec.to(device)
    for n, p in model.named_parameters():
        p.requires_grad_(n.startswith("audio_embed"))
    if config == "int8":
        quantize_linear_(model.layers, "int8", dynamic_int8_act=True)  # quantise, then adapt (train_metamathqa.py:178-179)
    if rank > 0:
        apply_linear_adapter_(model.layers, "lora", rank=rank, alpha=float(rank))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("lora_b"):
                p.normal_(0.0, 0.01, generator=g)  # non-zero so that every gradient path carries signal
    for n, p in model.named_parameters():
        if n.startswith("layers.") and n.endswith("_norm.weight"):
            p.requires_grad_(True)  # layer norms stay trainable as in the reference scripts
        if trainable == "reference" and n.startswith(("tok_embeddings.", "output.", "norm.")):
            p.requires_grad_(True)  # the reference's default: only model.layers is frozen / adapted (train_metamathqa.py:177-180)
    return model.train(), cfg


def cpu_baseline(seq: int, rank: int):
    """Oracle (CPU restatement of the reference) timed on the host cores on a bounded sample of the same workload."""
    from oracle import ref as O

    cores = max(1, min(_affinity_cores(), 64))  # a 1-GPU box shares its host: use the cores this process may run on
    torch.set_num_threads(cores)
    # sample: ONE of the 32 layers at the full sequence length (so attention's S^2 term is measured, not extrapolated) and the
    # LM head + CE on a quarter of the positions - about 10-30 s of CPU work on a 64-core host
    S = int(os.environ.get("LLX_CPU_BASELINE_SEQ", min(seq, 4096)))
    cfg = O.LLAMA31_8B._replace(num_layers=1, max_seq_len=S)
    D, I = cfg.embed_dim, cfg.intermediate_dim
    p = {}
    hq, hkv = cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim
    shapes = {"attention.wq": (hq, D), "attention.wk": (hkv, D), "attention.wv": (hkv, D), "attention.wo": (D, hq),
              "feed_forward.w1": (I, D), "feed_forward.w3": (I, D), "feed_forward.w2": (D, I)}
    gen = torch.Generator().manual_seed(0)
    for suf, (o, n) in shapes.items():
        p[f"layers.0.{suf}.weight"] = torch.randn(o, n, generator=gen) * 0.02
        p[f"layers.0.{suf}.lora_a"] = (torch.randn(rank, n, generator=gen) * 0.01).requires_grad_()
        p[f"layers.0.{suf}.lora_b"] = (torch.randn(o, rank, generator=gen) * 0.01).requires_grad_()
    p["layers.0.attention_norm.weight"] = torch.ones(D, requires_grad=True)
    p["layers.0.ffn_norm.weight"] = torch.ones(D, requires_grad=True)
    x = (torch.randn(1, S, D, generator=gen) * 0.5).requires_grad_()
    table = O.rope_table(cfg)
    t0 = time.perf_counter()
    y = O.layer(x, p, 0, cfg, table, None, 1.0)
    y.sum().backward()
    t_layer = time.perf_counter() - t0
    # LM head + CE on a slice of positions
    Sh = min(seq, 1024)
    w_out = torch.randn(128_256, D, generator=gen) * 0.02
    h = (torch.randn(1, Sh, D, generator=gen) * 0.5).requires_grad_()
    labels = torch.randint(0, 128_256, (1, Sh), generator=gen)
    t0 = time.perf_counter()
    loss = O.cross_entropy(torch.nn.functional.linear(O.rmsnorm(h, torch.ones(D)), w_out), labels)
    loss.backward()
    t_head = time.perf_counter() - t0
    step_s = 32 * t_layer * (seq / S) + t_head * (seq / Sh)  # x32 layers; linear in tokens where the sample is shorter than the step
    return {"value": round(seq / step_s, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32: 1 of 32 layers (8B dims, LoRA r={rank}) fwd+bwd at S={S} took {t_layer:.2f}s, LM head+CE on {Sh} positions "
                      f"{t_head:.2f}s; scaled to 32 layers and {seq} positions"}


def _affinity_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


WORKLOAD_TEXT = {
    "text": "Llama-3.1-8B text-only LoRA r={rank} bf16, seq={S}, 1 sequence per GPU, causal mask, base+LM head frozen (BASELINE.json configs[1]); "
            "random-init weights at 8B dimensions",
    "int8": "Llama-3.1-8B INT8 frozen base (dynamic int8 activations, i8 MFMA int8_mm_dequant) + bf16 LoRA r={rank}, seq={S}, 1 sequence per GPU "
            "(BASELINE.json configs[3] per-GPU workload)",
    "packed": "Llama-3.1-8B text-only LoRA r={rank} bf16, seq={S} packed from {n_docs} synthetic documents (log-normal lengths), document mask "
              "(BASELINE.json configs[1], packed variant of SURVEY 8d C2)",
    "audio": "Llama-3.1-8B + mel/Conv1D audio prefix ({St} audio tokens from {samples} samples) + {St} text tokens, prefix-LM mask, LoRA r={rank} + "
             "trainable audio_embed (BASELINE.json configs[2]; at seq 8192 the per-GPU shape of configs[4])",
}
PEAK = {"bf16": (2500.0, "TFLOP/s", "gemm_nt_kernel<EPI, false, 1> (bf16 MFMA GEMM, v_mfma_f32_16x16x32_bf16)"),
        "i8": (5000.0, "TOP/s", "gemm_nt_kernel<EPI, true, 1> (i8 MFMA GEMM = torchao::int8_mm_dequant, v_mfma_i32_16x16x64_i8)")}


def _traffic(args, config: str, kind: str, S: int):
    """HBM bytes per launch from the PMC passes committed under profiles/ FOR THIS workload (model, config, sequence length) and kernel
    (bench.py cannot run rocprofv3 on itself); None when no such file exists - a number measured on another workload is not a
    measurement of this one."""
    if args.model != "llama31_8b":
        return None
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{config}_s{S}_{kind}_gemm_hbm_traffic.json")))  # newest round last
    try:
        with open(files[-1]) as f:
            return round(json.load(f)["hbm_bytes_per_launch"])
    except (IndexError, OSError, KeyError, ValueError):
        return None


def _roofline(args, config: str, kind: str, st: dict, S: int) -> dict:
    peak, unit, kernel = PEAK[kind]
    ach = st["flops"] / (st["ms"] * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": _traffic(args, config, kind, S),
            "algorithmic_bytes_per_launch": round(st["alg_bytes"] / max(1, st["launches"])), "kernel": kernel,
            "launches_per_step": st["launches"], "avg_launch_us": round(st["ms"] * 1e3 / max(1, st["launches"]), 2),
            "gemm_ms_per_step": round(st["ms"], 2)}


def run_workload(args, config: str, device, world: int, rank: int, steps: int, warmup: int, seq: int | None = None, head_compact: bool | None = None) -> dict:
    """Build the model of one BASELINE configuration, capture its step, time `steps` steps after `warmup`, then trace one eager step with
    HIP events around every GEMM launch (bf16 and i8 kernels separately).  Returns the raw numbers; main() formats them."""
    from llx import kernels as K
    from llx import ops as llx_ops
    from llx.dp import GradBuckets

    head_compact_default = llx_ops._HEAD_COMPACT
    if head_compact is not None:
        llx_ops._HEAD_COMPACT = head_compact  # A/B of the labelled-rows-only LM head (llx/ops.py HeadLossFn); restored below
    S = seq or args.seq
    model, cfg = build_model(args.model, S, args.rank, device, config, args.trainable)
    trainable = [p for p in model.parameters() if p.requires_grad]
    if not args.no_arena:
        # LoRA factors + norm weights in one flat parameter / gradient arena: backward writes the gradients in place, AdamW is one
        # full-width launch, a data-parallel bucket is a slice (llx/arena.py)
        from llx.arena import TrainableArena

        trainable = TrainableArena(model).params()
    force_dp = os.environ.get("LLX_FORCE_DP") == "1"  # rehearse the N>1 code path (flat buckets + RCCL) on one GPU
    use_graph = not args.no_graph
    dp = world > 1 or force_dp
    optim = torch.optim.AdamW(trainable, lr=1e-4, weight_decay=0.0, fused=True, capturable=use_graph)

    gen = torch.Generator(device=device)
    gen.manual_seed(rank)  # rank-distinct data streams
    audio_cfg = config == "audio"
    St = S // 2 if audio_cfg else S  # audio: S/2 audio tokens (S/2 * 320 samples) + S/2 text tokens
    info = {"St": St, "samples": St * 320, "n_docs": 0}
    audio_buf = mask = None
    if audio_cfg:
        from modelling.llama import MaskSpec

        audio_buf = (torch.rand(1, St * 320, device=device, generator=gen) - 0.5) * 0.2
        mask = MaskSpec(prefix_len=torch.tensor([St], device=device, dtype=torch.int32))
    if config == "packed":
        # packed documents (train_metamathqa.py:51-83): lengths ~ clipped log-normal (median ~190, P99 ~680, max 2318 tokens as the
        # MetaMathQA statistics of SURVEY 8d), packed greedily into the S-token buffer; the unused tail keeps id 0 (packer quirk)
        import numpy as np
        from modelling.llama import MaskSpec

        rng = np.random.default_rng(1234 + rank)
        ids_np, pos, doc = np.zeros(S, dtype=np.int32), 0, 0
        while True:
            n = int(min(2318, max(16, rng.lognormal(mean=5.25, sigma=0.55))))
            if pos + n > S:
                break
            doc += 1
            ids_np[pos : pos + n] = doc
            pos += n
        info["n_docs"] = doc
        mask = MaskSpec(doc_ids=torch.from_numpy(ids_np).to(device))

    def batch():
        ids = torch.randint(0, cfg.vocab_size, (1, St), device=device, generator=gen)
        labels = torch.roll(ids, -1, 1)
        labels[:, : St // 4] = -100
        labels[:, -1] = -100
        return ids, labels

    def run_model(ids, labels):
        if audio_cfg:
            return model(audio_buf, ids, labels=labels, block_mask=mask)
        return model(ids, labels=labels, block_mask=mask)

    ids_buf, labels_buf = batch()
    info["labelled"] = int((labels_buf != -100).sum())  # positions that carry a label (the leading quarter is a masked prompt: SURVEY 8d C2)
    graph = opt_graph = static_loss = stepper = None
    if dp:
        # N > 1: forward / backward cut into 4 stages of 8 layers (llx.dp.StagedStep): the RCCL all-reduce of a stage's flat gradient
        # bucket runs under the backward of the next stage - from hipGraphs (one per stage) or eagerly (--no-graph)
        from llx.dp import StagedStep, llama_stages

        stages, stage_params = llama_stages(model, 4, labels=labels_buf, block_mask=mask, audio=audio_buf)
        stepper = StagedStep(model, stages, stage_params, optim, graph=use_graph, force=force_dp)
        buckets = stepper.buckets

        def eager_step():
            ids, labels = batch()
            ids_buf.copy_(ids)
            labels_buf.copy_(labels)
            return stepper._eager((ids_buf,))

        step = eager_step
        launch_mode = "eager, 4 backward stages, RCCL all-reduce of stage k under the backward of stage k-1"
        if use_graph:
            captured = True
            try:
                stepper.capture(ids_buf)
            except Exception as exc:  # noqa: BLE001 - a capture problem must not cost the measurement: the same stages run eagerly
                captured = False
                print(f"[bench] rank {rank}: stage capture failed ({type(exc).__name__}: {exc}); running the stages eagerly", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
            if world > 1:
                # graph or eager is decided by ALL ranks together: a rank replaying graphs next to one launching eagerly would still
                # exchange the same buckets, but the choice must not depend on which rank hit the problem
                flag = torch.tensor([1 if captured else 0], device=device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                captured = bool(flag.item())
            if captured:

                def step():
                    ids, labels = batch()
                    ids_buf.copy_(ids)
                    labels_buf.copy_(labels)
                    return stepper(ids_buf)

                launch_mode = "hipGraph replay per stage (fwd | 4 x bwd | AdamW), RCCL all-reduce of stage k under the backward of stage k-1"
            else:
                stepper._graphs = None
                step = eager_step
    else:
        buckets = GradBuckets(model, n_buckets=4, force=False, overlap=False)  # single replica: inactive, .grad stays with autograd

        def eager_step():
            ids, labels = batch()
            loss = run_model(ids, labels)
            loss.backward()
            buckets.finish()
            optim.step()
            buckets.zero_grad()
            return loss

        step = eager_step
        launch_mode = "eager"
        if use_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):  # warm caches (fused / transposed weight images, LDS attributes) outside the capture
                        buckets.zero_grad()
                        run_model(ids_buf, labels_buf).backward()
                        buckets.finish()
                        optim.step()
                torch.cuda.current_stream().wait_stream(side)
                buckets.zero_grad()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_loss = run_model(ids_buf, labels_buf)
                    static_loss.backward()
                    buckets.finish()
                    optim.step()

                def step():
                    ids, labels = batch()
                    ids_buf.copy_(ids)
                    labels_buf.copy_(labels)
                    graph.replay()
                    return static_loss

                launch_mode = "hipGraph replay (fwd+bwd+AdamW)"
            except Exception as exc:  # noqa: BLE001 - a capture problem must not cost the measurement: fall back to eager launches
                print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                optim = torch.optim.AdamW(trainable, lr=1e-4, weight_decay=0.0, fused=True)
                step = eager_step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- live roofline of the GEMM kernels: one more step with HIP events around every launch (on the launch stream).
    # Every rank runs the step (it contains the gradient exchange); only rank 0 records events.
    gemm_stats = {}
    attn_stats = {}
    # Three traced steps, the one with the median GEMM time is reported: a single eager step is one sample per launch, and an eager step
    # starts every kernel behind a host-side gap (clock and cache state differ from the graph replay's: 0.47-0.51 from run to run).
    traces = []
    for _ in range(3):
        if rank == 0:
            K.GEMM_TRACE = []
            K.ATTN_TRACE = []
        if not dp:
            buckets.zero_grad()
        eager_step()  # traced eagerly (events between launches), same kernels and shapes as the timed steps
        torch.cuda.synchronize()
        if rank == 0:
            traces.append((sum(e[0].elapsed_time(e[1]) for e in K.GEMM_TRACE), K.GEMM_TRACE, K.ATTN_TRACE))
            K.GEMM_TRACE = None
            K.ATTN_TRACE = None
    if rank == 0:
        _, tr, at = sorted(traces, key=lambda t: t[0])[1]
        for kind in ("fwd", "bwd"):
            sel = [e for e in at if e[2] == kind]
            if sel:
                attn_stats[kind] = {"calls": len(sel), "ms": sum(e[0].elapsed_time(e[1]) for e in sel), "B": sel[0][3], "S": sel[0][4], "H": sel[0][5]}
        for kind in ("bf16", "i8"):
            sel = [e for e in tr if e[4] == kind]
            if sel:
                # "launches" counts KERNEL launches (a GEMM whose last round would be half empty is two: full tiles + half tiles), as the
                # rocprofv3 kernel table and the PMC passes do
                gemm_stats[kind] = {"launches": sum(e[5] for e in sel), "calls": len(sel), "ms": sum(e[0].elapsed_time(e[1]) for e in sel),
                                    "flops": sum(e[2] for e in sel), "alg_bytes": sum(e[3] for e in sel)}
    res = {"config": config, "elapsed": elapsed, "steps": steps, "warmup": warmup, "per_step": per_step, "loss": float(loss.detach()),
           "launch": launch_mode, "gemm": gemm_stats, "attn": attn_stats, "info": info, "S": S}
    llx_ops._HEAD_COMPACT = head_compact_default
    # release the 16 GB of weights + cached images + graph pools before the next workload is built
    del step, eager_step, run_model, model, optim, buckets, trainable
    graph = opt_graph = static_loss = stepper = None  # noqa: F841
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    return res


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (a float4 copy reaches ~6.3 TB/s)


def run_decode(args, device, steps: int, warmup: int, contexts=(4096, 8192)) -> dict:
    """single-token decode of the 8B model against its KV cache (modelling/llama.py:76-90,126-127,189-207) - one hipGraph
    replay per token (embedding, 32 x [norm+q|k|v+RoPE+cache write, split-cache attention, wo+residual, norm+gate|up+SwiGLU, w2+residual],
    norm+head).  HBM-bound: every weight byte and every live K/V byte is read once per token; reported as ms/token and as the
    fraction of the 8 TB/s HBM peak those algorithmic bytes make of the measured time."""
    model, cfg = build_model(args.model, max(contexts), 0, device, "text")
    model.requires_grad_(False)
    model.eval()
    model.build_cache(inference=True)  # host-built tables (bit parity of the RoPE table), then moved with the caches
    model = model.to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(7)
    for layer in model.layers:  # a cache full of plausible keys / values (a real prefill of 8k tokens is not what is measured here)
        layer.attention.kv_cache.k_cache.normal_(0.0, 1.0, generator=gen)
        layer.attention.kv_cache.v_cache.normal_(0.0, 1.0, generator=gen)
    tok = torch.randint(0, cfg.vocab_size, (1, 1), device=device, generator=gen)
    pos = torch.zeros(1, dtype=torch.int64, device=device)
    weight_bytes = sum(p.numel() * p.element_size() for n, p in model.named_parameters() if not n.startswith("tok_embeddings")) + cfg.embed_dim * 2
    out = {"workload": f"Llama-3.1-8B bf16 single-token decode (batch 1) against a KV cache of max_seq_len {max(contexts)}; random-init weights, "
                       "random cache contents", "weight_bytes": weight_bytes, "launch": "hipGraph replay per token"}
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model(tok, input_pos=pos)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            logits = model(tok, input_pos=pos)
        for ctx in contexts:
            pos.fill_(ctx - 1)  # the token at position ctx-1 attends to ctx cached positions (its own k/v are written first)
            for _ in range(warmup):
                graph.replay()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            ev[0].record()
            for i in range(steps):
                graph.replay()
                ev[i + 1].record()
            torch.cuda.synchronize()
            per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
            ms = ev[0].elapsed_time(ev[steps]) / steps
            kv_bytes = cfg.num_layers * 2 * cfg.num_kv_heads * ctx * cfg.head_dim * 2
            gbs = (weight_bytes + kv_bytes) / (ms * 1e-3) / 1e9
            out[f"ctx{ctx}"] = {"ms_per_token": round(ms, 4), "p50_ms": round(per[len(per) // 2], 4), "tokens_per_s": round(1e3 / ms, 1), "kv_bytes": kv_bytes,
                                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                             "traffic": None, "algorithmic_bytes": weight_bytes + kv_bytes},
                                "finite_logits": bool(torch.isfinite(logits.float()).all())}
    del model, graph, logits
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    return out


def _summary(args, r: dict, world: int) -> dict:
    """Numbers of one workload: whole-job tokens/s, step times, the roofline of its dominant GEMM kernel (+ the i8 kernel's own when it ran)."""
    ms = r["elapsed"] / r["steps"] * 1e3
    ps = r["per_step"]
    d = {"value": round(r["S"] * world * r["steps"] / r["elapsed"], 1), "unit": "tokens/s", "steps": r["steps"], "warmup": r["warmup"],
         "ms_per_step": round(ms, 2), "p50_step_ms": round(ps[len(ps) // 2], 2), "p10_step_ms": round(ps[len(ps) // 10], 2),
         "p90_step_ms": round(ps[min(len(ps) - 1, (9 * len(ps)) // 10)], 2), "loss": round(r["loss"], 4), "launch": r["launch"],
         "workload": (WORKLOAD_TEXT[r["config"]].format(rank=args.rank, S=r["S"], **r["info"]) if args.model == "llama31_8b"
                      else f"tiny plumbing config seq={r['S']} ({r['config']})"),
         "labelled_positions": r["info"]["labelled"]}
    at = r.get("attn") or {}
    if "fwd" in at and "bwd" in at and r["config"] in ("text", "int8"):
        # attention kernels against the bf16 MFMA peak on ALGORITHMIC FLOPs: causal mask -> half of the S x S score matrix; forward
        # 2 products (QK^T, PV), backward 5 (S, dP, dV, dK, dQ - recomputation and whatever else the kernels execute is not counted)
        B_, S_, H_ = at["fwd"]["B"], at["fwd"]["S"], at["fwd"]["H"]
        per_product = 2.0 * B_ * H_ * S_ * S_ * 128 / 2
        fl = {"fwd": 2 * per_product * at["fwd"]["calls"], "bwd": 5 * per_product * at["bwd"]["calls"]}
        tot_ms = at["fwd"]["ms"] + at["bwd"]["ms"]
        ach = (fl["fwd"] + fl["bwd"]) / (tot_ms * 1e-3) / 1e12
        d["roofline_attn"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                              "traffic": None, "kernel": "attn_fwd_kernel + llx_attn_bwd (dQ [recomputes S, dP; publishes delta], dK/dV, reduce) - causal, algorithmic FLOPs (2 + 5 products; 7 executed in the backward)",
                              "fwd_us_per_layer": round(at["fwd"]["ms"] * 1e3 / at["fwd"]["calls"], 1), "bwd_us_per_layer": round(at["bwd"]["ms"] * 1e3 / at["bwd"]["calls"], 1),
                              "fwd_tflops": round(fl["fwd"] / (at["fwd"]["ms"] * 1e-3) / 1e12, 1), "bwd_tflops": round(fl["bwd"] / (at["bwd"]["ms"] * 1e-3) / 1e12, 1),
                              "attn_ms_per_step": round(tot_ms, 2)}
    g = {k: v for k, v in r["gemm"].items() if v["ms"] > 0}
    if g:
        dominant = max(g, key=lambda k: g[k]["ms"])  # the kernel the step spends most of its time in
        d["roofline"] = _roofline(args, r["config"], dominant, g[dominant], r["S"])
        if "i8" in g:
            d["roofline_i8"] = _roofline(args, r["config"], "i8", g["i8"], r["S"])  # int8_mm_dequant against the 5.0 POP/s i8 MFMA peak
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--model", default="llama31_8b", choices=["llama31_8b", "tiny"])
    ap.add_argument("--config", default="text", choices=["text", "int8", "audio", "packed", "decode"],
                    help="text: BASELINE configs[1] (headline); int8: configs[3] per-GPU (INT8 frozen base, dynamic int8 activations, i8 MFMA) ; "
                         "audio: configs[2] (mel+Conv1D prefix of seq/2 audio tokens + seq/2 text tokens, prefix-LM mask, audio_embed trainable)")
    ap.add_argument("--trainable", default="lora", choices=["lora", "reference"],
                    help="lora: adapters + layer norms (SURVEY 8d C2, the headline); reference: also tok_embeddings / norm / output, the reference "
                         "scripts' default trainable set (train_metamathqa.py:177-180) - weight gradients of the embedding and the LM head")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph (single GPU)")
    ap.add_argument("--no-arena", action="store_true", help="keep the trainable tensors separate (per-tensor AdamW launches; A/B of llx/arena.py)")
    ap.add_argument("--no-extras", action="store_true", help="skip the int8 / audio / packed workloads timed after the headline run (N=1, text only)")
    ap.add_argument("--extra-steps", type=int, default=6, help="timed steps of each extra workload (after 2 warm-up steps)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    # rehearsal of the N > 1 path on a one-GPU box: LLX_SINGLE_DEVICE=1 puts every rank on cuda:0 and LLX_DIST_BACKEND=gloo exchanges
    # through the host (RCCL refuses two ranks on one device); the measured path is one rank per GPU over RCCL ("nccl" on ROCm)
    if os.environ.get("LLX_SINGLE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("LLX_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or os.environ.get("LLX_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    S = args.seq
    if args.config == "decode":  # inference extra, its own line: ms per decoded token against the HBM roofline
        d = run_decode(args, device, args.steps, args.warmup)
        best = d["ctx4096"]
        print(json.dumps({"metric": "decode tokens/sec Llama-3.1-8B batch 1, KV cache 4096 (ms/token at 4096 and 8192)", "value": best["tokens_per_s"], "unit": "tokens/s",
                          "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": best["ms_per_token"], "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "config": {"workload": d["workload"]}, "roofline": best["roofline"], "decode": d}),
              flush=True)
        return
    r = run_workload(args, args.config, device, world, rank, args.steps, args.warmup)
    extras = {}
    if world == 1 and args.config == "text" and args.trainable == "lora" and not args.no_extras and os.environ.get("LLX_FORCE_DP") != "1":
        # the other single-GPU workloads of BASELINE.json, a few replays each in the same process (reported under "configs")
        # (+ the per-GPU shape of configs[4]: 4096 audio + 4096 text tokens, S = 8192, when the headline runs at its default length)
        for cfg, sq in (("int8", None), ("audio", None), ("packed", None)) + ((("audio_s8192", 8192),) if args.seq == 4096 and args.model == "llama31_8b" else ()):
            try:
                extras[cfg] = _summary(args, run_workload(args, cfg.split("_")[0], device, world, rank, args.extra_steps, 2, seq=sq), world)
            except Exception as exc:  # noqa: BLE001 - an extra workload must not cost the headline line
                extras[cfg] = {"error": f"{type(exc).__name__}: {exc}"}
                print(f"[bench] extra workload {cfg} failed: {exc}", file=sys.stderr, flush=True)
        try:  # the headline workload with the LM head over ALL rows (LLX_HEAD_COMPACT=0): the step time the labelled-row head is quoted against
            r_all = run_workload(args, "text", device, world, rank, args.extra_steps, 2, head_compact=False)
            extras["text_head_all_rows"] = {k: v for k, v in _summary(args, r_all, world).items() if k in ("value", "unit", "ms_per_step", "p50_step_ms", "loss", "steps")}
            extras["text_head_all_rows"]["lm_head_rows"] = "all"
        except Exception as exc:  # noqa: BLE001
            extras["text_head_all_rows"] = {"error": f"{type(exc).__name__}: {exc}"}
        if args.model == "llama31_8b":
            try:
                extras["decode"] = run_decode(args, device, 20, 5)
            except Exception as exc:  # noqa: BLE001
                extras["decode"] = {"error": f"{type(exc).__name__}: {exc}"}
                print(f"[bench] extra workload decode failed: {exc}", file=sys.stderr, flush=True)

    if rank == 0:
        sm = _summary(args, r, world)
        out = {
            "metric": "train tokens/sec Llama-3.1-8B seq4096 at 1/2/4/8 MI355X; p50 step ms",
            "value": sm["value"], "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sm["ms_per_step"], "p50_step_ms": sm["p50_step_ms"], "p10_step_ms": sm["p10_step_ms"], "p90_step_ms": sm["p90_step_ms"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": sm["workload"] + ("; tok_embeddings / norm / output trainable too (reference default)" if args.trainable == "reference" else ""),
                       "global_batch_tokens": S * world, "seq_len": S, "parallelism": f"dp{world}", "loss": sm["loss"],
                       "launch": sm["launch"],
                       # ignore_index positions (the masked prompt quarter) have no loss term and a zero gradient row: the LM head and the
                       # loss run over the labelled rows only unless LLX_HEAD_COMPACT=0 (bit-identical gradients, tests/test_model_gpu.py)
                       "labelled_positions_per_sequence": sm["labelled_positions"],
                       "lm_head_rows": "labelled only" if os.environ.get("LLX_HEAD_COMPACT", "1") != "0" and args.trainable == "lora" else "all"},
        }
        gf = GF_PER_TOKEN.get(S)
        if gf and args.model == "llama31_8b" and args.config == "text":
            # FLOPs the step REQUIRES: SURVEY 8d's 34.0 GF/token prices the frozen LM head (forward + d hidden = 4 x 525.3 M FLOPs per
            # position) on every position; ignore_index positions need neither (modelling/llama.py:216-218), so the head is priced on
            # the labelled positions only when the compacted head runs (all positions with LLX_HEAD_COMPACT=0 or a trainable head)
            head_gf = 4 * 4096 * 128_256 / 1e9
            rows = sm["labelled_positions"] if out["config"]["lm_head_rows"] == "labelled only" else S
            step_tf = ((gf - head_gf) * S + head_gf * rows) * 1e9 / 1e12
            out["step_mfma_frac"] = round(step_tf / (sm["ms_per_step"] * 1e-3) / BF16_DENSE_PEAK_TFLOPS, 4)
            out["step_algorithmic_tflop"] = round(step_tf, 2)
        for k in ("roofline", "roofline_i8", "roofline_attn"):
            if k in sm:
                out[k] = sm[k]
        if extras:
            out["configs"] = extras
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, args.rank)
        result_line = json.dumps(out)
    else:
        result_line = None
    if dist.is_initialized():
        dist.destroy_process_group()
    if result_line is not None:
        sys.stdout.flush()
        print(result_line, flush=True)  # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
