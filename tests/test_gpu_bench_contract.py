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


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` as ONE plain command (the driver's command shape): the parent starts the ranks itself
    through torch.distributed.run, touches no GPU, relays rank 0's line and exit code.  On a one-GPU box the two ranks
    share the card (peer-to-peer windows only: RCCL refuses two ranks on one device) -- a rehearsal of the protocol,
    not a measurement."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "p2p-only",
                          "--workload", "small", "--steps", "40", "--warmup", "4", "--repeats", "3", "--launch-timeout", "800"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["metric"] == "cg_iterations_per_sec" and j["n_gpus"] == 2 and j["steps"] == 40 and j["scaling"] == "strong"
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] / 1e3 - 1.0) < 1e-3
    assert j["config"]["partition"] == "1-D block rows x 2"
    d = j["diag"]
    for key in ("transport_used", "transports", "rccl_ranks", "devices_visible", "ranks_per_device", "cg_variant"):
        assert key in d, key
    assert d["rccl_ranks"] == 0 and d["transport_used"].startswith("p2p")
    picked = d["transports"]["picked"]
    assert "trial_ms_per_step" in d["transports"][picked], d["transports"]
    # the protocol modes run the same recurrence: their trial residuals agree (bench.py checks it, too)
    rz = [v["trial_rz"] for v in d["transports"].values() if isinstance(v, dict) and "trial_rz" in v]
    assert len(rz) >= 1 and all(abs(a - rz[0]) <= 1e-6 * abs(rz[0]) for a in rz)
