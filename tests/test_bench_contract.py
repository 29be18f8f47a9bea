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
