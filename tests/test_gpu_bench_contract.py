"""GPU: bench.py's one-line JSON contract on a small workload (the driver parses this line at round end)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "40",
                          "--warmup", "4", "--cpu-iters", "20", "--spmv-reps", "20"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["metric"] == "cg_iterations_per_sec" and j["n_gpus"] == 1 and j["steps"] == 40 and j["warmup"] == 4
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] / 1e3 - 1.0) < 1e-3
    assert j["dtype"] == "f64" and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "us_per_launch",
                "algorithmic_bytes_per_launch", "csr_equivalent"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes / measured launch time
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) <= 0.01 * r["achieved"] + 0.1
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "iterations/s" and c["value"] > 0 and 1 <= c["cores"] <= 16
