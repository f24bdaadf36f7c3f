"""bench.py's multi-rank bookkeeping WITHOUT a GPU: `CDA_BENCH_DRY_RUN=1` runs everything of an N-rank run that is not device work - the
launcher's rendezvous (gloo), the barriers, the MAX-over-ranks reduction, the command line's shape logic - and prints the line through the SAME builders
(bench.headline_line / bench.learner_line) from fabricated timings, so that the first real multi-GPU run cannot fail on formatting.  Nothing here is a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float, "higher_is_better": bool,
            "scaling": str, "dtype": str, "data": str, "config": dict}


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run(world, *flags):
    env = dict(os.environ, CDA_BENCH_DRY_RUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *flags]
    else:                                                            # the driver's own launch line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world), *flags]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                                # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    for k, t in CONTRACT.items():
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["unit"] == "agent-steps/s"
    assert d["n_gpus"] == world and "dry run" in d["data"]
    return d


def test_one_rank_line_has_the_contract_and_the_roofline_object():
    d = run(1, "--steps", "20", "--warmup", "5")
    assert d["steps"] == 20 and d["warmup"] == 5 and d["timed_repeats"]["n"] == 5 and d["config"]["collective"] == "none"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["concurrent_launches"] == 2 and r["markets_per_launch"] == 2048 and r["algorithmic_bytes_per_launch"] == 2048 * (1444 + 4 * 324)
    assert "4096 markets x 4" in d["config"]["workload"] and "configs[2]" in d["config"]["workload"]


@pytest.mark.parametrize("transport", ["torch", "rccl"])
def test_two_ranks_hand_back_line(transport):
    d = run(2, "--steps", "20", "--warmup", "5", "--transport", transport)
    cf = d["config"]
    assert "2 all-gathers per step" in cf["collective"] and "208-B records" in cf["collective"] and f"asked: {transport}" in cf["collective"]
    assert ("ncclAllGather" if transport == "rccl" else "torch.distributed") in cf["collective"]
    assert "global 8192 markets" in cf["workload"] and cf["groups"] == 2 and d["roofline"]["kernel"] == "k_step"
    # value = the units ALL ranks processed / the slowest rank's time (rank 1 fabricates +1 %): 2 x 4096 x 4 x 20 / (20 x 40.4 us)
    assert abs(d["value"] - 2 * 4096 * 4 / 40.4e-6) / d["value"] < 1e-9 and abs(d["ms_per_step"] - 0.0404) < 1e-9


def test_two_ranks_data_parallel_learner_line():
    d = run(2, "--learner", "dp")
    assert d["steps"] == 8 and d["warmup"] == 2 and "all-reduce of the 0.9-MB gradient" in d["config"]["collective"] and "global 8192" in d["config"]["workload"]
    assert abs(d["value"] - 2 * 4096 * 4 * 64 / 7.07e-3) / d["value"] < 1e-9 and set(d["loss_last_iteration"]) == {"pg_loss", "v_loss", "entropy"}


def test_c4_shape_and_no_gather_flag():
    d = run(2, "--config", "c4", "--steps", "300", "--warmup", "10", "--no-gather")
    assert d["config"]["markets_per_gpu"] == 2048 and d["config"]["agents"] == 8 and d["config"]["collective"] == "none" and d["config"]["groups"] == 4
    assert d["timed_repeats"]["n"] == 1 and "configs[3]" in d["config"]["workload"]
