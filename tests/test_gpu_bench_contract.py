"""bench.py's contract with the driver, on a small workload: stdout is ONE JSON line with the agreed
keys (metric / value / ... / roofline / cpu_baseline), also when it starts its own ranks
(--gpus 2 over gloo on the one card of the test box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--rows", "200000", "--k", "100", "--max-total", "50", "--steps", "2", "--warmup", "1",
         "--settle-seconds", "0", "--phase-steps", "1", "--cpu-sample-rows", "20000"]


def _run(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_one_json_line_with_the_agreed_keys():
    d = _run([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "rows/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "strong" and d["data"] == "synthetic" and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 200000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "rows/s" and c["value"] > 0
    assert c["gpu_vs_oracle_max_rel_alpha_err"] <= 1e-9 and "sample" in c
    assert d["parity_fragile"] in (True, False)


def test_two_ranks_started_by_the_bench_itself():
    d = _run(["--gpus", "2", "--backend", "gloo", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["rows_total"] == 200000 and d["config"]["rows_per_gpu"] == 100000
    assert set(d["phases_ms"]["untimed_steps_with_events"]) >= {"stats", "allreduce"}
    w = _run(["--gpus", "2", "--backend", "gloo", "--no-cpu-baseline", "--scaling", "weak"])
    assert w["scaling"] == "weak" and w["config"]["rows_total"] == 400000
