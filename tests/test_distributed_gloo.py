"""CPU, world_size 2, gloo: the N>1 path - contiguous market sharding, global-index seeding, the packed all-gather of
obs/reward/flags and the hand-back of compact per-market records with the receiving side's rebuild of the stacked
observation (gym_continuousdoubleauction_amd/parallel.py).  The local stepper is injected (the CPU oracle stands in for the
HIP env, which needs a GPU) and so is the receiving side (a numpy restatement of cda_handback_unpack; the kernel itself is
checked against the same restatement on the GPU, tests/test_hip_facade.py); the check is that the global outputs every rank
ends up with are identical to one process stepping all markets - i.e. independent of the shard count."""
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
    """CDAVecEnv-shaped stand-in over the CPU oracle (torch CPU tensors).  With `groups` it also keeps the hand-back records the
    product's kernels write (include/cda.h cda_set_handback: f32 frame[42] | f64 reward[A] | terminated, truncated, restarted)."""

    def __init__(self, config, n, device, groups=0):
        import oracle_lib as O
        from gym_continuousdoubleauction_amd.parallel import handback_stride
        self.o = O.OracleEnv(config, n_markets=n)
        self.obs_dim, self.num_agents = self.o.obs_dim, self.o.A
        self.handback = torch.zeros((n, handback_stride(self.num_agents)), dtype=torch.uint8)
        g = max(1, groups)
        self.group_ranges = [(n * k // g, n * (k + 1) // g - n * k // g) for k in range(g)]

    def _record(self, obs, rew, term, trunc, restarted):
        hb, a = self.handback.numpy(), self.num_agents
        hb[:, :168] = np.ascontiguousarray(obs[:, -42:]).view(np.uint8)
        hb[:, 168:168 + 8 * a] = np.ascontiguousarray(rew).view(np.uint8)
        hb[:, 168 + 8 * a] = term; hb[:, 169 + 8 * a] = trunc; hb[:, 170 + 8 * a] = restarted

    def reset(self, seed):
        obs = self.o.reset(seeds=seed.numpy().astype(np.uint64)).copy()
        n = obs.shape[0]
        self._record(obs, np.zeros((n, self.num_agents)), np.zeros(n, np.uint8), np.zeros(n, np.uint8), 1)
        return torch.from_numpy(obs)

    def step(self, cat, mean, sigma, price, off, present=None, pipelined=False):
        obs, rew, term, trunc, _ = self.o.step(cat.numpy(), mean.numpy(), sigma.numpy(), price.numpy(), off.numpy())
        self._record(obs, rew, term, trunc, 0)
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


def unpack_restated(records, n_segments, seg_records, seg_row_stride, row0, num_agents, n_hist, obs, reward, term, trunc, n_rows_total=0):
    """cda_handback_unpack restated in numpy (the receiving side of the hand-back): shift the row by one frame, append the new one
    (restarted: every frame = the new one), overwrite reward and flags.  n_rows_total > 0 (uneven shards): segment s owns seg_row_stride
    rows, the last one the remainder; records beyond a segment's rows are padding and are skipped."""
    rec = records.numpy().reshape(n_segments * seg_records, -1)
    o, r, te, tr = obs.numpy(), reward.numpy(), term.numpy(), trunc.numpy()
    a = num_agents
    for j in range(rec.shape[0]):
        seg, local = j // seg_records, row0 + j % seg_records
        if n_rows_total > 0 and local >= (n_rows_total - seg * seg_row_stride if seg == n_segments - 1 else seg_row_stride):
            continue
        row = seg * seg_row_stride + local
        frame = rec[j, :168].view(np.float32)
        if rec[j, 170 + 8 * a]:
            o[row] = np.tile(frame, n_hist)
        else:
            o[row, :-42] = o[row, 42:].copy()
            o[row, -42:] = frame
        r[row] = rec[j, 168:168 + 8 * a].view(np.float64)
        te[row], tr[row] = rec[j, 168 + 8 * a], rec[j, 169 + 8 * a]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_continuousdoubleauction_amd.parallel import ShardedVecEnv
    # three group chains per rank, each with its own communicator and its own collective per step
    env = ShardedVecEnv(CFG, N_TOTAL, device=None, env_factory=lambda c, n, d, g: _OracleStepper(c, n, d, g), groups=3, handback=True,
                        unpack=unpack_restated)
    assert (env.first, env.n_local) == (rank * N_TOTAL // world, N_TOTAL // world) and len(env.group_ranges) == 3
    obs0 = env.reset(seed_base=1000)
    assert np.array_equal(env.full[0][env.first:env.first + env.n_local].numpy().view(np.uint32), obs0.numpy().view(np.uint32))
    rec = []
    for t in range(T):
        acts = [x[env.first:env.first + env.n_local] for x in _actions(t)]
        obs, rew, term, trunc, _ = env.step(*acts)                # ... the hand-back records travel, `full` is rebuilt on every rank
        g = env.gather(obs, rew, term, trunc)                     # the packed gather of whole observations says the same
        rec.append([x.clone().numpy() for x in g])
        fo, fr, ft, fu = env.full
        assert np.array_equal(fo.numpy().view(np.uint32), g[0].numpy().view(np.uint32)), (rank, t)
        assert np.array_equal(fr.numpy().view(np.uint64), g[1].contiguous().numpy().view(np.uint64)), (rank, t)
        assert np.array_equal(ft.numpy() != 0, g[2].numpy()) and np.array_equal(fu.numpy() != 0, g[3].numpy())
    # a reset after the run: its records carry `restarted`, and every rank's full observation restarts from them
    o = env.reset(seed_base=2000)
    z = torch.zeros(env.n_local, dtype=torch.bool)
    g = env.gather(o, torch.zeros(env.n_local, A, dtype=torch.float64), z, z)
    assert np.array_equal(env.full[0].numpy().view(np.uint32), g[0].numpy().view(np.uint32))
    assert not env.full[2].any() and not env.full[3].any() and not env.full[1].any()
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
    env = ShardedVecEnv(CFG, n_total, device=None, env_factory=lambda c, n, d, g: _OracleStepper(c, n, d))
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
    # the hand-back with uneven shards: every rank steps and sends the largest shard's record count (the smaller shards padded with markets
    # nobody reads), the receiving side skips the padding; `full` equals the packed gather on every rank, every step
    hb = ShardedVecEnv(CFG, n_total, device=None, env_factory=lambda c, n, d, g: _OracleStepper(c, n, d, g), groups=2, handback=True, unpack=unpack_restated)
    assert hb.n_env == n_total // world + n_total % world and hb.n_local == shard_range(rank, world, n_total)[1] and hb.uneven
    o0 = hb.reset(seed_base=1000)
    assert o0.shape[0] == hb.n_local
    for t in range(8):
        rng = np.random.default_rng(500 + t)
        full = (rng.integers(0, 9, (n_total, A)).astype(np.int32), rng.uniform(-1, 1, (n_total, A)).astype(np.float32),
                rng.uniform(0, 1, (n_total, A)).astype(np.float32), rng.integers(0, 10, (n_total, A)).astype(np.int32),
                rng.integers(0, 3, (n_total, A)).astype(np.int32))
        acts = [torch.from_numpy(x[hb.first:hb.first + hb.n_local]) for x in full]
        obs, rew, term, trunc, _ = hb.step(*acts)
        assert obs.shape[0] == hb.n_local
        assert np.array_equal(hb.full[0].numpy().view(np.uint32), rec[t][0].view(np.uint32)), (rank, t)
        assert np.array_equal(hb.full[1].numpy().view(np.uint64), np.ascontiguousarray(rec[t][1]).view(np.uint64)), (rank, t)
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


def _dp_worker(rank, world, port, outdir):
    """the data-parallel learner's rule with the PyTorch statement of the network: a rank's loss is normalised by the GLOBAL minibatch, the gradients are summed"""
    import torch
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gym_continuousdoubleauction_amd import ppo
    from gym_continuousdoubleauction_amd.parallel import make_grad_allreduce, shard_range
    torch.manual_seed(5)                                      # every rank builds the same network
    m = ppo.ActorCritic(168)
    g = torch.Generator().manual_seed(11)                     # ... and knows the whole synthetic batch; it works on its own rows
    R, A = 96, 4
    x = torch.randn(R, 168, generator=g)
    acts = (torch.randint(0, 9, (R * A,), generator=g), torch.randint(0, 10, (R * A,), generator=g), torch.randint(0, 3, (R * A,), generator=g), torch.randn(R * A, 2, generator=g))
    lp_old, adv, ret = torch.randn(R * A, generator=g) * 0.1 - 7, torch.randn(R * A, generator=g), torch.randn(R * A, generator=g)

    def loss_of(rows, norm_rows):
        sel = (torch.arange(A)[None, :] + rows[:, None] * A).reshape(-1)
        logp, ent, v = m.evaluate(x[rows], tuple(a[sel] for a in acts), agents_per_row=A)
        ratio = (logp - lp_old[sel]).exp()
        per = -torch.min(ratio * adv[sel], ratio.clamp(0.8, 1.2) * adv[sel]) + 0.5 * (v - ret[sel]).pow(2) - 0.01 * ent
        return per.sum() / (norm_rows * A)
    first, cnt = shard_range(rank, world, R)
    loss_of(torch.arange(first, first + cnt), R).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    allreduce = make_grad_allreduce(dist)
    allreduce(flat)
    for p in m.parameters():
        p.grad = None
    loss_of(torch.arange(R), R).backward()                     # the single-process gradient on the union batch
    want = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.allclose(flat, want, rtol=1e-4, atol=1e-7), float((flat - want).abs().max())
    # the advantage sums of a rollout travel the same way
    s = torch.tensor([float(adv[first * A:(first + cnt) * A].sum()), float((adv[first * A:(first + cnt) * A] ** 2).sum())], dtype=torch.float64)
    allreduce(s)
    assert abs(float(s[0]) - float(adv.sum())) < 1e-4 and abs(float(s[1]) - float((adv ** 2).sum())) < 1e-4
    if rank == 0:
        open(os.path.join(outdir, "dp_ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_learner_rule_two_ranks(tmp_path):
    """every rank back-propagates its own shard with the loss normalised by the global minibatch; ONE all-reduce (sum) of the gradient gives the
    single-process gradient on the union batch (here with the PyTorch statement of the network; the HIP kernels under the same rule: tests/test_hip_dp.py)."""
    port = 29500 + (os.getpid() * 7 + 3) % 400
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "dp_ok").exists()
