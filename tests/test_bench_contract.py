"""GPU: bench.py prints ONE JSON line with the contract's fields (metric/value/..., roofline, cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "8", "--markets", "512",
                                   "--cpu-seconds", "0.5"], cwd=ROOT, stderr=subprocess.DEVNULL, text=True)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 8 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "agent-steps/s" and d["value"] > 1e6
    assert abs(d["value"] - 512 * 4 * 40 / (d["ms_per_step"] * 1e-3 * 40)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "traffic" in r and r["kernel"] == "k_step" and r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "agent-steps/s" and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
