"""bench.py contract: one JSON line with the required keys (tiny plumbing configs on the GPU; the CPU test only checks the CLI)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline"}


def test_cli_help():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--config", "int8"], ["--config", "audio", "--no-graph"], ["--config", "packed"]])
def test_tiny_bench_line(cuda, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--model", "tiny", "--seq", "512", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = out.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert REQUIRED <= d.keys(), REQUIRED - d.keys()
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] in ("TFLOP/s", "TOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["unit"] == "TFLOP/s" or "--config" in extra and "int8" in extra  # only the int8 workload may be dominated by the i8 kernel
    if "int8" in extra:
        assert d["roofline_i8"]["peak"] == 5000.0 and d["roofline_i8"]["unit"] == "TOP/s" and d["roofline_i8"]["traffic"] is None
    assert r["traffic"] is None  # no PMC file exists for a tiny plumbing workload
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.gpu
def test_two_rank_bench_rehearsal_on_one_gpu(cuda):
    """The default `--gpus N` path (StagedStep: per-stage hipGraphs, bucket exchange between them) with TWO ranks, launched exactly as the
    driver launches it.  A one-GPU box cannot give each rank its own device and RCCL refuses two ranks on one device, so both ranks sit
    on cuda:0 (LLX_SINGLE_DEVICE=1) and exchange through gloo (LLX_DIST_BACKEND=gloo): plumbing, not performance."""
    env = dict(os.environ, LLX_SINGLE_DEVICE="1", LLX_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--seq", "512", "--steps", "3", "--warmup", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch_tokens"] == 1024
    assert "RCCL all-reduce of stage k under the backward of stage k-1" in d["config"]["launch"] and "cpu_baseline" not in d and "configs" not in d
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
