"""CPU, world_size 2, gloo: the N>1 path - contiguous market sharding, global-index seeding and the
packed all-gather of obs/reward/flags (gym_continuousdoubleauction_amd/parallel.py).  The local stepper is
injected (the CPU oracle stands in for the HIP env, which needs a GPU); the check is that the gathered
global outputs are identical to one process stepping all markets - i.e. independent of the shard count."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL, A, T = 12, 4, 24
CFG = {"num_of_agents": A, "init_cash": 1000000, "max_step": 64, "is_render": False}


class _OracleStepper:
    """CDAVecEnv-shaped stand-in over the CPU oracle (torch CPU tensors)."""

    def __init__(self, config, n, device):
        import oracle_lib as O
        self.o = O.OracleEnv(config, n_markets=n)
        self.obs_dim, self.num_agents = self.o.obs_dim, self.o.A

    def reset(self, seed):
        return torch.from_numpy(self.o.reset(seeds=seed.numpy().astype(np.uint64)).copy())

    def step(self, cat, mean, sigma, price, off, present=None):
        obs, rew, term, trunc, _ = self.o.step(cat.numpy(), mean.numpy(), sigma.numpy(), price.numpy(), off.numpy())
        return (torch.from_numpy(obs.copy()), torch.from_numpy(rew.copy()), torch.from_numpy(term.astype(bool)),
                torch.from_numpy(trunc.astype(bool)), {})

    def close(self):
        self.o.close()


def _actions(t):
    rng = np.random.default_rng(100 + t)
    return (torch.from_numpy(rng.integers(0, 9, (N_TOTAL, A)).astype(np.int32)),
            torch.from_numpy(rng.uniform(-1, 1, (N_TOTAL, A)).astype(np.float32)),
            torch.from_numpy(rng.uniform(0, 1, (N_TOTAL, A)).astype(np.float32)),
            torch.from_numpy(rng.integers(0, 10, (N_TOTAL, A)).astype(np.int32)),
            torch.from_numpy(rng.integers(0, 3, (N_TOTAL, A)).astype(np.int32)))


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_continuousdoubleauction_amd.parallel import ShardedVecEnv
    env = ShardedVecEnv(CFG, N_TOTAL, device=None, env_factory=lambda c, n, d: _OracleStepper(c, n, d))
    assert (env.first, env.n_local) == (rank * N_TOTAL // world, N_TOTAL // world)
    env.reset(seed_base=1000)
    rec, rec_async, prev = [], [], None

    def _drain(h):
        o, r, te, tr = h.wait()            # per-rank views [world, n_local, ...]
        rec_async.append([o.reshape(N_TOTAL, -1).clone().numpy(), r.reshape(N_TOTAL, -1).clone().numpy(),
                          te.reshape(-1).clone().numpy(), tr.reshape(-1).clone().numpy()])

    for t in range(T):
        acts = [x[env.first:env.first + env.n_local] for x in _actions(t)]
        obs, rew, term, trunc, _ = env.step(*acts)
        g = env.gather(obs, rew, term, trunc)
        rec.append([x.clone().numpy() for x in g])
        h = env.gather_async(outputs=(obs, rew, term, trunc))     # the pipelined loop of bench.py: <= 2 in flight
        if prev is not None:
            _drain(prev)
        prev = h
    _drain(prev)
    for a, b in zip(rec, rec_async):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8))
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"), obs=np.stack([r[0] for r in rec]), rew=np.stack([r[1] for r in rec]),
                 term=np.stack([r[2] for r in rec]), trunc=np.stack([r[3] for r in rec]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_uneven(rank, world, port, outdir, n_total):
    """13 markets over 2 ranks: 6 + 7 (the last rank takes the remainder); the packed gather pads and trims."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_continuousdoubleauction_amd.parallel import ShardedVecEnv, shard_range
    env = ShardedVecEnv(CFG, n_total, device=None, env_factory=lambda c, n, d: _OracleStepper(c, n, d))
    assert (env.first, env.n_local) == shard_range(rank, world, n_total) == (rank * (n_total // world), n_total // world + (n_total % world if rank == world - 1 else 0))
    env.reset(seed_base=1000)
    rec = []
    for t in range(8):
        rng = np.random.default_rng(500 + t)
        full = (rng.integers(0, 9, (n_total, A)).astype(np.int32), rng.uniform(-1, 1, (n_total, A)).astype(np.float32),
                rng.uniform(0, 1, (n_total, A)).astype(np.float32), rng.integers(0, 10, (n_total, A)).astype(np.int32),
                rng.integers(0, 3, (n_total, A)).astype(np.int32))
        acts = [torch.from_numpy(x[env.first:env.first + env.n_local]) for x in full]
        g = env.gather(*env.step(*acts)[:4])
        assert g[0].shape[0] == n_total
        rec.append([x.clone().numpy() for x in g])
    with pytest.raises(ValueError):
        env.gather_async(outputs=env.step(*acts)[:4])
    if rank == 0:
        np.savez(os.path.join(outdir, "uneven.npz"), obs=np.stack([r[0] for r in rec]), rew=np.stack([r[1] for r in rec]))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_last_rank_takes_the_remainder(tmp_path):
    n_total = 13
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_uneven, args=(2, port, str(tmp_path), n_total), nprocs=2, join=True)
    got = np.load(tmp_path / "uneven.npz")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    ref = _OracleStepper(CFG, n_total, None)
    ref.reset(torch.arange(1000, 1000 + n_total, dtype=torch.int64))
    for t in range(8):
        rng = np.random.default_rng(500 + t)
        full = (rng.integers(0, 9, (n_total, A)).astype(np.int32), rng.uniform(-1, 1, (n_total, A)).astype(np.float32),
                rng.uniform(0, 1, (n_total, A)).astype(np.float32), rng.integers(0, 10, (n_total, A)).astype(np.int32),
                rng.integers(0, 3, (n_total, A)).astype(np.int32))
        obs, rew, _, _, _ = ref.step(*[torch.from_numpy(x) for x in full])
        assert np.array_equal(got["obs"][t].view(np.uint32), obs.numpy().view(np.uint32)), t
        assert np.array_equal(got["rew"][t].view(np.uint64), rew.numpy().view(np.uint64)), t


def test_two_rank_shards_gather_to_the_single_process_result(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    ref = _OracleStepper(CFG, N_TOTAL, None)
    ref.reset(torch.arange(1000, 1000 + N_TOTAL, dtype=torch.int64))
    for t in range(T):
        obs, rew, term, trunc, _ = ref.step(*_actions(t))
        assert np.array_equal(got["obs"][t].view(np.uint32), obs.numpy().view(np.uint32)), t
        assert np.array_equal(got["rew"][t].view(np.uint64), rew.numpy().view(np.uint64)), t
        assert np.array_equal(got["term"][t], term.numpy()) and np.array_equal(got["trunc"][t], trunc.numpy())
