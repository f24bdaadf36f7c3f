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


def test_bench_multi_gpu_code_path_on_one_rank():
    """The N>1 path of bench.py (RCCL process group, double-buffered output slabs, asynchronous all-gather
    overlapped with the next launch) with a single rank - what the driver launches under torchrun at N>1."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    for extra in ([], ["--no-overlap"]):
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--markets",
                                       "512", "--no-cpu-baseline", "--force-gather"] + extra, cwd=ROOT, env=env,
                                      stderr=subprocess.STDOUT, text=True, timeout=600)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 1 and d["value"] > 1e6 and "all_gather" in d["config"]["collective"]
        assert d["config"]["flagged_markets"] == 0 and d["roofline"]["kernel_ms"] > 0


def test_bench_fused_mode_reports_the_same_contract():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "64", "--warmup", "32", "--markets", "512",
                                   "--no-cpu-baseline", "--fused", "32"], cwd=ROOT, stderr=subprocess.DEVNULL, text=True, timeout=600)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["steps"] == 64 and "FUSED" in d["config"]["workload"] and d["roofline"]["traffic"] is None and d["value"] > 1e6
    assert d["config"]["flagged_markets"] == 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "32", "--markets", "64", "--no-cpu-baseline",
                          "--fused", "32"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "multiples" in (bad.stderr + bad.stdout)
