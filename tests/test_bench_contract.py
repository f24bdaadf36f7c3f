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


def test_bench_two_ranks_on_one_gpu():
    """Two processes under torch.distributed.run, both pinned to GPU 0, gloo instead of RCCL: the multi-rank logic of bench.py
    (global-index seeds per rank, slab all-gather sized by the world, schedule calibration with its all-reduce, max-over-ranks
    timing) runs for real; only the transport differs from the driver's 2-GPU launch."""
    env = dict(os.environ, CDA_BENCH_DEVICE="0", CDA_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "48",
                          "--warmup", "16", "--markets", "256", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 48 and "all_gather" in d["config"]["collective"] and d["config"]["flagged_markets"] == 0
    assert "calibrated" in (d["config"]["gather_schedule"] or "") and "global 512 markets" in d["config"]["workload"]
    assert abs(d["value"] - 2 * 256 * 4 * 48 / (d["ms_per_step"] * 1e-3 * 48)) / d["value"] < 1e-6
