"""bench.py's one-line contract (the driver parses it): metric / value / unit / step counts, the roofline object of the
dominant kernel measured live, the CPU baseline of the same run, and the fields that must NOT claim more than was measured."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout       # exactly ONE json line on stdout
    return json.loads(lines[0])


def test_training_bench_line():
    d = run_bench("--gpus", "1", "--steps", "8", "--warmup", "2")
    assert d["metric"].startswith("227x227 RGB tiles/sec fwd+bwd") and d["unit"] == "tiles/s"
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]     # value = tiles of all ranks / time
    assert 5000 < d["value"] < 60000
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "tiles/s" and 1 <= c["cores"] <= 16 and 0 < c["value"] < d["value"] and c["sample"]


def test_inference_bench_line():
    d = run_bench("--mode", "infer", "--image-side", "2048", "--steps", "3", "--warmup", "1")
    assert d["unit"] == "Mpx/s" and d["n_gpus"] == 1 and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["config"]["exchanged_pixels"] == 0 and d["value"] > 100
