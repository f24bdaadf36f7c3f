"""GPU: the data-parallel learner ("if the learner is itself data-parallel over the same shards, the all-gather can be skipped
entirely").  Two processes on ONE GPU over gloo (RCCL refuses two ranks on one device: tests/test_bench_contract.py) run the real kernels: each rank's fused
update on its half of a batch with the loss normalised by the global minibatch, ONE all-reduce of the gradient per step, the same Adam step on every rank -
against a single process stepping through the union batch.  Unmeasured on more than one GPU (no node)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, A, STEPS, LR = 1024, 4, 3, 1e-3


def _batch():
    from gym_continuousdoubleauction_amd import mlp
    g = torch.Generator().manual_seed(6)
    th = mlp.init_theta(generator=torch.Generator().manual_seed(13))
    x = torch.randn(R, 168, generator=g) * 0.75
    rec = torch.zeros(R, A, 8)
    rec[..., 0] = torch.randint(0, 9, (R, A), generator=g).int().view(torch.float32)
    rec[..., 1] = torch.randint(0, 10, (R, A), generator=g).int().view(torch.float32)
    rec[..., 2] = torch.randint(0, 3, (R, A), generator=g).int().view(torch.float32)
    rec[..., 3:5] = torch.randn(R, A, 2, generator=g)
    rec[..., 5] = torch.randn(R, A, generator=g) * 0.1 - 7.0
    rec[..., 6:8] = torch.randn(R, A, 2, generator=g)
    return th, x, rec


def _steps(th, x, rec, rows, world, allreduce):
    """STEPS optimiser steps over the rows [rows) in natural order (identity permutation: the union of the ranks' minibatches is the single process's)"""
    from gym_continuousdoubleauction_amd import mlp
    p = mlp.FusedPolicy("cuda:0", theta=th)
    n = rows.stop - rows.start
    upd = mlp.FusedUpdate(p, n, n, A, chunks=4, allreduce=allreduce, world=world)
    xd, recd = x[rows].contiguous().cuda(), rec[rows].contiguous().cuda()
    perms = torch.arange(n).repeat(STEPS, 1).cuda()
    upd.run(xd, epochs=STEPS, clip=0.2, vf_coef=0.5, ent_coef=0.01, lr=LR, max_norm=0.5, perms=perms, records=(recd, None, 0))
    torch.cuda.synchronize()
    return p, upd


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from gym_continuousdoubleauction_amd.parallel import make_grad_allreduce, shard_range
    th, x, rec = _batch()
    first, cnt = shard_range(rank, world, R)
    p, upd = _steps(th, x, rec, slice(first, first + cnt), world, make_grad_allreduce(dist))
    torch.save({"theta": p.theta.cpu(), "grad": upd.grad.cpu(), "norm2": upd.norm2[2].cpu(), "step": p.adam_step.cpu(), "out6": upd.out6.cpu()}, os.path.join(outdir, f"dp_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_take_the_single_process_steps(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, 30700 + os.getpid() % 300, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(tmp_path / f"dp_{r}.pt") for r in range(2)]
    th, x, rec = _batch()
    p, upd = _steps(th, x, rec, slice(0, R), 1, None)
    want = p.theta.cpu()
    assert torch.equal(got[0]["theta"], got[1]["theta"]) and float(got[0]["step"]) == STEPS      # the ranks stay in lockstep
    # the summed gradient of the last step is the union batch's (float32 sums in another order), and so is its norm (recomputed after the all-reduce)
    g, gw = got[0]["grad"].double(), upd.grad.cpu().double()
    assert (g - gw).norm() <= 2e-4 * gw.norm(), float((g - gw).norm() / gw.norm())
    assert abs(float(got[0]["norm2"]) - float(upd.norm2[2])) <= 1e-3 * float(upd.norm2[2])
    # the loss statistics travel with the gradient: every rank holds the GLOBAL means (what adapt_kl_coef and the logs read), not its share of them
    o, ow = got[0]["out6"].double(), upd.out6.cpu().double()
    assert torch.equal(got[0]["out6"], got[1]["out6"]) and float(ow[:4].abs().min()) > 0
    assert ((o[:4] - ow[:4]).abs() <= 2e-3 * ow[:4].abs() + 1e-5).all(), (o, ow)
    moved = (want - th).double().norm()                          # (Adam normalises every coordinate's step: a coordinate whose gradient is at rounding level may step the other way - norms, not maxima)
    assert float((want - th).abs().max()) > 0.5 * LR and (got[0]["theta"] - want).double().norm() <= 0.02 * moved, (float(moved), float((got[0]["theta"] - want).double().norm()))


def test_bench_learner_dp_two_ranks_on_one_gpu():
    """bench.py --learner dp under torch.distributed.run, both ranks pinned to GPU 0, gloo instead of RCCL: the data-parallel PPO loop end to end (global-index
    seeds per shard, gradient + advantage-sum all-reduces, max-over-ranks timing); `config.collective` names what travelled."""
    env = dict(os.environ, CDA_BENCH_DEVICE="0", CDA_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(29300 + os.getpid() % 150), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--learner", "dp", "--steps", "3",
                          "--warmup", "1", "--markets", "256"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-2500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and "all-reduce of the 0.9-MB gradient" in d["config"]["collective"] and "gloo" in d["config"]["collective"]
    assert d["config"]["flagged_markets"] == 0 and d["value"] > 1e5 and "global 512" in d["config"]["workload"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--learner", "dp", "--steps", "3", "--warmup", "1", "--markets", "256"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert one.returncode == 0 and d1["n_gpus"] == 1 and d1["config"]["collective"] == "none (one rank)"


def _league_worker(rank, world, port, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd.league_train import train_league_fused
    from gym_continuousdoubleauction_amd.parallel import make_grad_allreduce
    N = 128
    env = CDAVecEnv({"num_of_agents": 8, "init_cash": 1000000, "max_step": 16, "is_render": False, "auto_reset": True}, n_markets=N, with_info=False)
    bank, league, hist = train_league_fused(env, iters=4, horizon=16, num_trainable=2, std_dev_multiplier=-10.0, min_iterations_between_champions=2, log=lambda s: None,
                                            allreduce=make_grad_allreduce(dist), world=world, first_market=rank * N, chains=2)
    torch.cuda.synchronize()
    torch.save({"theta": bank.theta[:2].cpu(), "champions": [c["id"] for c in league.history], "returns": [h.get("module_returns") for h in hist],
                "slot_net": bank.slot_net.cpu(), "flags": int((env.flags() != 0).sum())}, os.path.join(outdir, f"league_{rank}.pt"))
    dist.barrier()
    env.close()
    dist.destroy_process_group()


def test_league_data_parallel_ranks_stay_in_lockstep(tmp_path):
    """The league under the data-parallel learner (two ranks, one GPU, gloo): both ranks hold the SAME two policies after four iterations (summed gradients, global
    advantage statistics), see the same module returns (summed over the shards) and promote the same champions; their shards' opponents differ (global episode ids)."""
    import torch.multiprocessing as mp
    mp.spawn(_league_worker, args=(2, 31200 + os.getpid() % 300, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / f"league_{r}.pt") for r in range(2))
    assert torch.equal(a["theta"], b["theta"]) and not torch.equal(a["theta"][0], a["theta"][1])
    assert a["champions"] == b["champions"] and len(a["champions"]) == 2
    assert a["returns"] == b["returns"] and a["returns"][0] is not None
    assert not torch.equal(a["slot_net"], b["slot_net"]) and a["flags"] == b["flags"] == 0
