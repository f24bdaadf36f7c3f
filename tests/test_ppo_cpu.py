"""CPU: the PPO loop's math and plumbing on a stand-in env (the CPU oracle behind the CDAVecEnv surface)."""
import numpy as np
import torch

from gym_continuousdoubleauction_amd import ppo


class _CpuEnv:
    """CDAVecEnv-shaped stand-in over the CPU oracle (test infrastructure only)."""

    def __init__(self, n, a, max_step):
        import oracle_lib as O
        self.o = O.OracleEnv({"num_of_agents": a, "init_cash": 1000000, "max_step": max_step, "is_render": False}, n)
        self.n_markets, self.num_agents, self.obs_dim = n, a, self.o.obs_dim
        self.obs = torch.zeros((n, self.obs_dim))

    def reset(self, seed=None, mask=None):
        seeds = None if seed is None else (np.arange(self.n_markets, dtype=np.uint64) + np.uint64(seed))
        m = None if mask is None else mask.numpy().astype(np.uint8)
        self.obs.copy_(torch.from_numpy(self.o.reset(seeds=seeds, mask=m)))
        return self.obs

    def step(self, cat, mean, sigma, price, off):
        obs, rew, term, trunc, _ = self.o.step(cat.numpy(), mean.numpy(), sigma.numpy(), price.numpy(), off.numpy())
        self.obs.copy_(torch.from_numpy(obs))
        return self.obs, torch.from_numpy(rew.copy()), torch.from_numpy(term.astype(bool)), torch.from_numpy(trunc.astype(bool)), {}


def test_gae_matches_the_textbook_recursion():
    rew = torch.tensor([[1.0], [0.0], [2.0]]); val = torch.tensor([[0.5], [0.4], [0.3]]); done = torch.tensor([[0.0], [0.0], [1.0]])
    adv, ret = ppo.gae(rew, val, torch.tensor([9.0]), done, gamma=0.9, lam=0.8)
    d2 = 2.0 - 0.3
    d1 = 0.0 + 0.9 * 0.3 - 0.4
    d0 = 1.0 + 0.9 * 0.4 - 0.5
    a2 = d2; a1 = d1 + 0.9 * 0.8 * a2; a0 = d0 + 0.9 * 0.8 * a1
    assert torch.allclose(adv.squeeze(), torch.tensor([a0, a1, a2]), atol=1e-6)
    assert torch.allclose(ret, adv + val)


def test_action_mapping_respects_the_space_bounds():
    torch.manual_seed(0)
    m = ppo.ActorCritic(168)
    acts, logp, val = m.act(torch.randn(40, 168))
    cat, mean, sigma, price, off = ppo.to_env_actions(acts, 10, 4)
    assert cat.shape == (10, 4) and cat.dtype == torch.int32 and 0 <= int(cat.min()) and int(cat.max()) <= 8
    assert float(mean.min()) >= -1 and float(mean.max()) <= 1 and float(sigma.min()) >= 0 and float(sigma.max()) <= 1
    assert int(price.max()) <= 9 and int(off.max()) <= 2 and logp.shape == (40,) and val.shape == (40,)


def test_ppo_loop_runs_and_resets_truncated_markets():
    env = _CpuEnv(6, 3, max_step=5)
    model, hist = ppo.train(env, iters=2, horizon=12, log=lambda s: None)
    assert len(hist) == 2 and all(np.isfinite(h["pg_loss"]) and np.isfinite(h["v_loss"]) for h in hist)
    assert hist[0]["agent_steps"] == 6 * 3 * 12


def test_league_self_play_loop_grows_a_champion_pool():
    from gym_continuousdoubleauction_amd.league_train import train_league

    class _Env(_CpuEnv):
        def __init__(self, n, a, max_step):
            super().__init__(n, a, max_step)
            self.max_step = max_step
    env = _Env(8, 4, max_step=6)
    model, mapper, hist = train_league(env, iters=3, num_trainable=1, promote_margin=-1e9, log=lambda s: None)
    assert len(hist) == 3 and all(np.isfinite(h["pg_loss"]) and np.isfinite(h["episode_return"]) for h in hist)
    assert hist[0]["promoted"] == "champion_1" and mapper.available_modules[:4] == ["policy_0", "policy_1", "policy_2", "policy_3"]
    assert [n for n in mapper.available_modules if n.startswith("champion_")] == ["champion_1", "champion_2", "champion_3"]
    assert hist[-1]["pool"][-1] == "champion_3"


def test_split_k_weight_gradient_and_closed_form_evaluate_equal_the_plain_formulation():
    """The update's two restructurings are numerically the plain ones: (i) a Linear whose weight gradient is a split-K batched
    product (ppo._SplitKLinear) gives the gradients of nn.Linear; (ii) evaluate()'s closed-form log-probability / entropy equal
    the torch.distributions objects act() samples from."""
    torch.manual_seed(0)
    n = 64 * ppo._SplitKLinear.SPLIT
    x = torch.randn(n, 24, dtype=torch.float64, requires_grad=True)
    lin = ppo._Linear(24, 10).double()
    ref = torch.nn.Linear(24, 10).double()
    ref.load_state_dict(lin.state_dict())
    g = torch.randn(n, 10, dtype=torch.float64)
    (lin(x) * g).sum().backward()
    gx, gw, gb = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    x.grad = None
    (ref(x) * g).sum().backward()
    assert torch.allclose(gx, x.grad) and torch.allclose(gw, ref.weight.grad, rtol=1e-10, atol=1e-10) and torch.allclose(gb, ref.bias.grad)
    m = ppo.ActorCritic(168).double()
    obs = torch.randn(300, 168, dtype=torch.float64)
    acts, logp_a, val_a = m.act(obs)
    logp_e, ent_e, val_e = m.evaluate(obs, acts)
    cat, price, off, cont = m.dists(obs)
    ent = cat.entropy() + price.entropy() + off.entropy() + cont.entropy().sum(-1)
    assert torch.allclose(logp_e.double(), logp_a.double(), atol=1e-5) and torch.allclose(ent_e.double(), ent.double(), atol=1e-5) and torch.allclose(val_e.double(), val_a.double(), atol=1e-6)


def test_shared_observation_update_equals_the_per_sample_update():
    """`agents_per_row = A`: the network sees each market-step's observation ONCE and the row's outputs serve its A samples.  One
    full-batch epoch of that update must move the parameters exactly as the plain update on the A-times replicated rows does (same
    samples, same loss, gradients summed per row instead of per replica).  float64 network; the loss side runs in float32 in both
    (evaluate() casts the logits), so the two parameter steps agree to float32 rounding of the gradient sums: 1e-5 of the step."""
    import copy
    torch.manual_seed(3)
    R, A = 96, 4
    base = ppo.ActorCritic(168).double()
    obs = torch.randn(R, 168, dtype=torch.float64)
    with torch.no_grad():
        acts, logp_old, _ = base.act(obs.repeat_interleave(A, dim=0))
        logp_old = logp_old + 0.2 * torch.randn_like(logp_old)
    adv, ret = torch.randn(R * A, dtype=torch.float64), torch.randn(R * A, dtype=torch.float64)
    results = []
    for per_row in (A, 1):
        m = copy.deepcopy(base)
        opt = torch.optim.SGD(m.parameters(), lr=0.1)
        x = obs if per_row == A else obs.repeat_interleave(A, dim=0)
        st = ppo.ppo_update(m, opt, x, acts, logp_old, adv, ret, epochs=1, minibatch=R * A, fused=False, agents_per_row=per_row)
        results.append((m, st))
    (m_shared, s_shared), (m_plain, s_plain) = results
    for k in s_plain:
        assert abs(s_shared[k] - s_plain[k]) < 1e-6 * max(1.0, abs(s_plain[k])), (k, s_shared, s_plain)     # (the statistics pass through float32)
    moved = 0.0
    with torch.no_grad():
        for (n1, p1), (_, p2), (_, p0) in zip(m_shared.named_parameters(), m_plain.named_parameters(), base.named_parameters()):
            step = float((p2 - p0).abs().max())
            assert float((p1 - p2).abs().max()) <= 1e-5 * step + 1e-12, (n1, float((p1 - p2).abs().max()), step)
            moved += float((p1 - p0).abs().sum())
    assert moved > 1e-3                                                  # (the step was not a no-op)


def test_ppo_loop_per_sample_forward_still_runs():
    env = _CpuEnv(4, 3, max_step=5)
    model, hist = ppo.train(env, iters=1, horizon=8, log=lambda s: None, shared_obs=False)
    assert len(hist) == 1 and np.isfinite(hist[0]["pg_loss"]) and hist[0]["agent_steps"] == 4 * 3 * 8


def test_block_structure_of_the_actor_critic_survives_updates():
    """Policy and value networks are stored as block matrices whose off-block entries start at zero and get no gradient: after real
    optimizer steps they are still exactly zero (the value head never reads policy units and vice versa), and the block network
    equals two separate 256x256 MLPs built from its blocks."""
    torch.manual_seed(4)
    m = ppo.ActorCritic(168).double()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    R, A = 64, 2
    obs = torch.randn(R, 168, dtype=torch.float64)
    with torch.no_grad():
        acts, logp_old, _ = m.act(obs.repeat_interleave(A, dim=0))
    adv, ret = torch.randn(R * A, dtype=torch.float64), torch.randn(R * A, dtype=torch.float64)
    w2_before = m.l2.weight.detach().clone()
    ppo.ppo_update(m, opt, obs, acts, logp_old, adv, ret, epochs=3, minibatch=64, fused=False, agents_per_row=A)
    H = m.hidden
    w2, wo = m.l2.weight.detach(), m.out.weight.detach()
    assert float((w2 - w2_before).abs().max()) > 1e-4                                    # (the steps were real)
    assert float(w2[:H, H:].abs().max()) == 0.0 and float(w2[H:, :H].abs().max()) == 0.0
    assert float(wo[:m.N_OUT, H:].abs().max()) == 0.0 and float(wo[m.N_OUT, :H].abs().max()) == 0.0 and float(wo[m.N_OUT + 1:].abs().max()) == 0.0
    with torch.no_grad():
        o, v = m.trunk(obs)
        h1 = torch.tanh(obs @ m.l1.weight.t() + m.l1.bias)
        hp = torch.tanh(h1[:, :H] @ w2[:H, :H].t() + m.l2.bias[:H])
        hv = torch.tanh(h1[:, H:] @ w2[H:, H:].t() + m.l2.bias[H:])
        o_sep = hp @ wo[:m.N_OUT, :H].t() + m.out.bias[:m.N_OUT]
        v_sep = hv @ wo[m.N_OUT, H:] + m.out.bias[m.N_OUT]
    assert torch.allclose(o, o_sep, atol=1e-12) and torch.allclose(v, v_sep, atol=1e-12)
