"""GPU: the two BASELINE configurations round 1 only exercised in miniature.

configs[3]  16 384 markets x 8 agents over 8 GPUs: (a) the per-GPU shard, 2 048 x 8 for 256 steps, HIP vs the CPU oracle
            on every 16th market + flags + NAV conservation over all markets; (b) the whole 16 384 x 8 batch on one GPU
            (it fits: 102 MB) checked through size-independent properties.
configs[4]  4 096 x 4 driven by a PyTorch-ROCm PPO policy: one iteration with horizon 64; the action tensors the policy
            produced are replayed through the oracle on 32 sampled markets.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config3_per_gpu_shard_2048x8_matches_oracle_on_every_16th_market():
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    import oracle_lib as O
    n, a, steps, stride = 2048, 8, 256, 16
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 1024, "is_render": False}
    env = CDAVecEnv(cfg, n, groups=2)
    sample = np.arange(0, n, stride)
    ora = O.OracleEnv(cfg, len(sample))
    first = 3 * n                                              # rank 3 of the 8-GPU job: global markets 6144 .. 8191
    seeds = np.arange(1000 + first, 1000 + first + n, dtype=np.uint64)
    o = env.reset(seed=seeds).cpu().numpy()
    assert np.array_equal(o[sample].view(np.uint32), ora.reset(seeds[sample]).view(np.uint32))
    for t in range(steps):
        acts = env.random_actions(t, action_seed=2024, market_index_base=first)
        obs, rew, term, trunc, info = env.step(*acts)
        env.join()
        oo, orw, ot, otr, oi = ora.step(*[x[sample] for x in acts])
        assert np.array_equal(obs.cpu().numpy()[sample].view(np.uint32), oo.view(np.uint32)), t
        assert np.array_equal(rew.cpu().numpy()[sample].view(np.uint64), orw.view(np.uint64)), t
        if t % 32 == 31:
            nav = info["nav"].cpu().numpy()[sample].reshape(len(sample), a, 16)
            assert np.array_equal(nav, oi["nav"].view(np.uint8).reshape(len(sample), a, 16)), t
    for j in range(0, len(sample), 8):
        assert bytes(env.get_state(int(sample[j]))) == bytes(ora.get_state(j)), j
    assert (env.flags() == 0).all()
    err, bad = env.nav_conservation()
    assert not bad.any() and float(err.max()) < 1e-6
    assert (env.check_invariants() == 0).all()
    assert int(env.book_peak().max()) < 256
    env.close(); ora.close()


def test_config3_whole_batch_16384x8_properties():
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    n, a, steps = 16384, 8, 192
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 1024, "is_render": False}
    env = CDAVecEnv(cfg, n, groups=2)
    env.reset(seed=1000)
    acts = env.random_actions_device(0, steps, action_seed=2024)
    for t in range(steps):
        obs, rew, term, trunc, info = env.step(*[x[t] for x in acts])
    env.join()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and not term.any() and not trunc.any()
    flags = env.flags()
    assert (flags == 0).all(), torch.bincount(flags.flatten().to(torch.int64))
    inv = env.check_invariants()                              # sorted, uncrossed, positive, escrow identity, positions net to zero
    assert (inv == 0).all(), torch.unique(inv)
    assert (info["net_position"].sum(dim=1) == 0).all()
    err, bad = env.nav_conservation()
    assert not bad.any() and float(err.max()) < 1e-6
    peak = env.book_peak()
    assert 0 < int(peak.max()) < 256
    # determinism + independence of the batch: market 12345 alone replays to the same state
    solo = CDAVecEnv(cfg, 1)
    solo.reset(seed=np.array([1000 + 12345], np.uint64))
    for t in range(steps):
        host = env.random_actions(t, action_seed=2024)
        solo.step(*[x[12345:12346] for x in host])
    assert bytes(solo.get_state(0)) == bytes(env.get_state(12345))
    env.close(); solo.close()


def test_config4_ppo_iteration_at_4096x4_with_oracle_replay():
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    import oracle_lib as O
    n, a, horizon = 4096, 4, 64
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n, with_info=False)
    sample = torch.arange(0, n, n // 32, device="cuda:0")
    rec = []

    def hook(it, step, env_acts, obs, rew, term, trunc):
        rec.append(tuple(x[sample].cpu().numpy() for x in env_acts) + (obs[sample].cpu().numpy(), rew[sample].cpu().numpy()))

    logs = []
    model, hist = ppo.train(env, iters=1, horizon=horizon, seed=11, log=logs.append, rollout_hook=hook)
    assert not any("capture failed" in x for x in logs), logs
    h = hist[0]
    assert all(math.isfinite(h[k]) for k in ("pg_loss", "v_loss", "entropy", "mean_reward")) and h["agent_steps"] == n * a * horizon
    assert (env.flags() == 0).all() and (env.check_invariants() == 0).all()
    err, bad = env.nav_conservation()
    assert not bad.any()
    # the policy's own action tensors, replayed on the CPU: same rewards and observations, bit for bit
    assert len(rec) == horizon
    idx = sample.cpu().numpy()
    ora = O.OracleEnv({k: v for k, v in cfg.items() if k != "auto_reset"}, len(idx))
    ora.reset(seeds=(11 + idx).astype(np.uint64))             # env.reset(seed=11): market i is seeded 11 + i
    for t, (cat, mean, sigma, price, off, obs, rew) in enumerate(rec):
        assert cat.min() >= 0 and cat.max() <= 8 and np.abs(mean).max() <= 1 and sigma.min() >= 0 and sigma.max() <= 1
        oo, orw, _, _, _ = ora.step(cat, mean, sigma, price, off)
        assert np.array_equal(rew.view(np.uint64), orw.view(np.uint64)), t
        assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32)), t
    env.close(); ora.close()
    del model
    torch.cuda.synchronize()


def test_config4_fused_ppo_iteration_at_4096x4_with_oracle_replay():
    """BASELINE configs[4] on the hand-written network (ppo.train_fused): one iteration at full size - four rollout chains x 64 steps, 16
    minibatch steps of 262 144 samples.  The rollout's OWN buffers are the evidence: the recorded actions of 32 markets replayed through the CPU
    oracle reproduce the recorded observations and rewards bit for bit; losses finite, no flags, invariants and NAV conservation hold; the
    update moved the parameters and the first minibatch step's loss statistics are those of an on-policy batch (entropy of a fresh policy)."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    import oracle_lib as O
    n, a, horizon = 4096, 4, 64
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n, with_info=False)
    keep, logs = {}, []
    pol, hist = ppo.train_fused(env, iters=1, horizon=horizon, seed=11, log=logs.append, keep=keep)
    h = hist[0]
    assert all(math.isfinite(h[k]) for k in ("pg_loss", "v_loss", "entropy", "mean_reward")) and h["agent_steps"] == n * a * horizon
    assert 6.0 < h["entropy"] < 8.0                              # log 9 + log 10 + log 3 + two Gaussians at log_std -0.5: 7.4 for a fresh policy
    assert float(pol.adam_step.item()) == 16 and torch.isfinite(pol.theta).all()
    assert (env.flags() == 0).all() and (env.check_invariants() == 0).all()
    _, bad = env.nav_conservation()
    assert not bad.any()
    b = {k: v.cpu().numpy() for k, v in keep["buffers"].items()}
    idx = np.arange(0, n, n // 32)
    ora = O.OracleEnv({k: v for k, v in cfg.items() if k != "auto_reset"}, len(idx))
    o0 = ora.reset(seeds=(11 + idx).astype(np.uint64))            # env.reset(seed=11): market i is seeded 11 + i
    assert np.array_equal(b["obs"][0][idx].view(np.uint32), o0.view(np.uint32))
    for t in range(horizon):
        cat, mean, sigma = b["category"][t][idx], b["size_mean"][t][idx], b["size_sigma"][t][idx]
        assert cat.min() >= 0 and cat.max() <= 8 and np.abs(mean).max() <= 1 and sigma.min() >= 0 and sigma.max() <= 1
        oo, orw, _, _, _ = ora.step(cat, mean, sigma, b["price"][t][idx], b["price_offset"][t][idx])
        assert np.array_equal(b["reward"][t][idx].view(np.uint64), orw.view(np.uint64)), t
        assert np.array_equal(b["obs"][t + 1][idx].view(np.uint32), oo.view(np.uint32)), t
    env.close(); ora.close()
    del pol
    torch.cuda.synchronize()
