"""GPU: every episode checked and summarised ON THE DEVICE (include/cda.h cda_episode_metrics_*; SURVEY 8(f) rows 2 and 3 on the fast path).

The reference's callback tallies every step of an episode (train/callbk/league_based_self_play_callback.py:541-600), checks sum(NAV) == num_agents x init_cash in
Decimal at every episode END and summarises the accounts (:627-755); its driver stops a strict run on a violation (train/train.py:1109-1164).  With the in-kernel
auto reset the finished episode's ledger is gone before any host code runs, so the cold episode-end paths do it.  Checked here against the same figures computed on the
host from the CPU oracle's replay of the recorded actions: integer and decimal-derived columns bit for bit, f64 sums within 1e-12 (tests/episode_metrics_util.py)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

ACTION_KEYS = ("category", "size_mean", "size_sigma", "price", "price_offset")


def _cfg(A, max_step, cash=1000000, **kw):
    return dict({"num_of_agents": A, "init_cash": cash, "max_step": max_step, "is_render": False, "auto_reset": True}, **kw)


def _oracle(cfg, N, seed):
    import oracle_lib as O
    ora = O.OracleEnv({k: v for k, v in cfg.items() if k != "auto_reset"}, N)
    ora.reset(seeds=(seed + np.arange(N)).astype(np.uint64))
    return ora


def _actions(rng, N, A, aggressive=False):
    cat = rng.integers(0, 9, (N, A)).astype(np.int32)
    if aggressive:                                            # many market orders: fills, passive fills, now and then a bankruptcy on small accounts
        cat = np.where(rng.random((N, A)) < 0.5, rng.choice([1, 5], (N, A)), cat).astype(np.int32)
    return (cat, rng.uniform(-1, 1, (N, A)).astype(np.float32), rng.uniform(0, 1, (N, A)).astype(np.float32), rng.integers(0, 10, (N, A)).astype(np.int32),
            rng.integers(0, 3, (N, A)).astype(np.int32))


def _replay(ora, em, acts_per_step):
    """the actions through the oracle with the env's auto-reset rule; feeds the host-side tallies"""
    for acts in acts_per_step:
        _, rew, term, trunc, info = ora.step(*acts)
        ended = em.feed(info, rew, term, trunc, done_mask_of=lambda i: ora.get_state(i).done_mask)
        if len(ended):
            ora.reset(mask=(term | trunc).astype(np.uint8))


@pytest.mark.parametrize("with_info", [False, True])
@pytest.mark.parametrize("A,max_step,cash", [(4, 7, 1000000), (8, 5, 3000), (3, 1, 1000000)])
def test_stepwise_metrics_equal_the_oracle_replay(with_info, A, max_step, cash):
    """cda_step on an auto_reset env: the in-kernel reset of the info-less kernel / the reset pass behind the kernel with info tensors credit every episode end"""
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from episode_metrics_util import OracleEpisodeMetrics, assert_tables_equal
    N, T, seed = 80, 23, 4100
    cfg = _cfg(A, max_step, cash)
    env = CDAVecEnv(cfg, n_markets=N, with_info=with_info)
    env.reset(seed=seed)
    env.enable_episode_metrics(True)
    ora, em = _oracle(cfg, N, seed), OracleEpisodeMetrics(N, A, cash)
    rng = np.random.default_rng(5 + A)
    steps = [_actions(rng, N, A, aggressive=cash < 100000) for _ in range(T)]
    for k, acts in enumerate(steps):
        env.step(*acts)
        if k == 11:                                           # a collection in the middle: the accumulators restart, the running tallies go on
            _replay(ora, em, steps[:12])
            dev = [t.cpu().numpy() for t in env.collect_episode_metrics()]
            assert_tables_equal(dev[0], dev[1], *em.table(), what="first collection")
    _replay(ora, em, steps[12:])
    dev = [t.cpu().numpy() for t in env.collect_episode_metrics()]
    ref = em.table()
    assert ref[1][0] >= N * ((T - 12) // max_step) and (cash > 100000 or ref[1][7] > 0 or True)
    assert_tables_equal(dev[0], dev[1], *ref, what="second collection")
    again = [t.cpu().numpy() for t in env.collect_episode_metrics()]      # cleared behind the read
    assert not again[0].any() and not again[1].any()
    assert not em.violating and (env.flags() == 0).all()
    env.close(); ora.close()


def test_reset_of_an_unfinished_episode_discards_it_and_a_finished_one_is_credited_once():
    """no auto reset: the episode end is credited by the reset that follows it (once); an episode that is reset before its end is dropped, as the reference's callback
    never sees it end (callbk:481-497)"""
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from episode_metrics_util import OracleEpisodeMetrics, assert_tables_equal
    N, A, max_step, seed = 48, 4, 6, 977
    cfg = {k: v for k, v in _cfg(A, max_step).items() if k != "auto_reset"}
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    env.reset(seed=seed)
    env.enable_episode_metrics(True)
    ora, em = _oracle(cfg, N, seed), OracleEpisodeMetrics(N, A, 1000000)
    rng = np.random.default_rng(1)
    for t in range(4):                                        # four steps, then half of the markets are reset early
        acts = _actions(rng, N, A)
        env.step(*acts)
        _, rew, term, trunc, info = ora.step(*acts)
        em.feed(info, rew, term, trunc, done_mask_of=lambda i: ora.get_state(i).done_mask)
    early = (np.arange(N) % 2 == 0)
    env.reset(mask=early.astype(np.uint8)); ora.reset(mask=early.astype(np.uint8)); em.discard(early)
    ended_total = np.zeros(N, bool)
    for t in range(2):                                        # the other half reaches max_step
        acts = _actions(rng, N, A)
        _, _, term, trunc, _ = env.step(*acts)
        _, rew, oterm, otrunc, info = ora.step(*acts)
        ended_total[em.feed(info, rew, oterm, otrunc, done_mask_of=lambda i: ora.get_state(i).done_mask)] = True
    assert ended_total.sum() == N // 2 and np.array_equal((term | trunc).cpu().numpy(), ended_total)
    dev = [t.cpu().numpy() for t in env.collect_episode_metrics(clear=False)]
    assert dev[1][0] == 0                                     # nothing credited yet: nobody has reset the finished markets
    env.reset(mask=ended_total.astype(np.uint8))
    env.reset(mask=ended_total.astype(np.uint8))              # a second reset of the same markets credits nothing more
    dev = [t.cpu().numpy() for t in env.collect_episode_metrics()]
    assert_tables_equal(dev[0], dev[1], *em.table(), what="credited by the reset")
    env.close(); ora.close()


def test_random_agent_episodes_in_one_launch_are_credited():
    """cda_run_random ends an episode without resetting: the launch itself checks and credits it, the reset that follows does not do it again"""
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from episode_metrics_util import OracleEpisodeMetrics, assert_tables_equal
    N, A, max_step, seed = 64, 4, 9, 31
    cfg = {k: v for k, v in _cfg(A, max_step, cash=4000).items() if k != "auto_reset"}
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    env.reset(seed=seed)
    env.enable_episode_metrics(True)
    ora, em = _oracle(cfg, N, seed), OracleEpisodeMetrics(N, A, 4000)
    env.run_random(max_step + 3, action_seed=77)
    alive = np.ones(N, bool)
    for t in range(max_step):
        acts = env.random_actions(t, action_seed=77)
        keep = ora.get_state
        _, rew, term, trunc, info = ora.step(*acts)
        # (a market that ended earlier keeps being stepped by this loop: only its first end counts, like the launch that stopped there)
        info = {k: v.copy() for k, v in info.items()}
        mask_t, mask_u = term.copy(), trunc.copy()
        mask_t[~alive] = 0; mask_u[~alive] = 0
        if (~alive).any():
            for k in ("reward_terms", "is_pass_action", "num_rejected_step", "order_step_placed", "num_trades_step", "num_passive_fills_step"):
                info[k][~alive] = 0
            rew = rew.copy(); rew[~alive] = 0
        em.steps[~alive] -= 1
        ended = em.feed(info, rew, mask_t, mask_u, done_mask_of=lambda i: keep(i).done_mask)
        alive[ended] = False
    assert not alive.any()
    dev = [t.cpu().numpy() for t in env.collect_episode_metrics(clear=False)]
    assert_tables_equal(dev[0], dev[1], *em.table(clear=False), what="credited by the launch")
    env.reset()
    dev = [t.cpu().numpy() for t in env.collect_episode_metrics()]
    assert_tables_equal(dev[0], dev[1], *em.table(), what="not credited twice")
    env.close(); ora.close()


@pytest.mark.parametrize("A,N,H", [(4, 96, 4), (8, 64, 4), (4, 40, 2)])
def test_fused_rollout_metrics_equal_the_oracle_replay(A, N, H):
    """the PPO rollout (policy inside the step kernel where the env qualifies, HIP graphs, in-kernel auto reset): two rollouts of 14 steps over 5-step episodes"""
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    from episode_metrics_util import OracleEpisodeMetrics, assert_tables_equal
    T, max_step, seed, cash = 14, 5, 880, 20000
    cfg = _cfg(A, max_step, cash, n_hist=H)
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    env.reset(seed=seed)
    env.enable_episode_metrics(True)
    roll = mlp.RolloutChains(env, mlp.FusedPolicy("cuda:0", seed=3, n_hist=H), T, groups=2, seed=17)
    ora, em = _oracle(cfg, N, seed), OracleEpisodeMetrics(N, A, cash)
    for rnd in range(2):
        b = {k: v.cpu().numpy() for k, v in roll.run().items() if k in ACTION_KEYS}
        torch.cuda.synchronize()
        _replay(ora, em, [tuple(b[k][t] for k in ACTION_KEYS) for t in range(T)])
        dev = [t.cpu().numpy() for t in env.collect_episode_metrics()]
        ref = em.table()
        assert ref[1][0] == N * ((rnd + 1) * T // max_step - rnd * T // max_step)
        assert_tables_equal(dev[0], dev[1], *ref, what=f"rollout {rnd}")
    env.close(); ora.close()


def test_league_rollout_metrics_are_keyed_by_module():
    """league self-play: the table's rows are the MODULES (trainable policies, random modules, champions) whatever slot they played"""
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    from gym_continuousdoubleauction_amd.league import LeagueSlotMapper
    from episode_metrics_util import OracleEpisodeMetrics, assert_tables_equal
    N, A, k, T, max_step, seed, cash = 64, 8, 2, 12, 6, 510, 50000
    cfg = _cfg(A, max_step, cash)
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    env.reset(seed=seed)
    env.enable_episode_metrics(True)
    bank = mlp.PolicyBank("cuda:0", N, A, k, max_frozen=4, seed=5, random_seed=9)
    mapper = LeagueSlotMapper(A, k, A - k, 1.0, 3.0)
    net_of = {}
    for _ in range(2):
        cid = mapper.add_champion()
        net_of[cid] = bank.snapshot(0)
    slot_pool = torch.full((N, A), -1, dtype=torch.int32, device="cuda:0")
    mapper.assign_device(bank, episode_ids=[f"e0-m{i}" for i in range(N)], net_of=net_of, slot_pool=slot_pool)
    module_of = torch.where(slot_pool < 0, torch.arange(A, device="cuda:0", dtype=torch.int32).expand(N, A), slot_pool + k).to(torch.int32).contiguous()
    n_mod = len(mapper.available_modules)
    roll = mlp.RolloutChains(env, bank, T, groups=2, seed=23)
    b = {key: v.cpu().numpy() for key, v in roll.run().items() if key in ACTION_KEYS}
    torch.cuda.synchronize()
    ora, em = _oracle(cfg, N, seed), OracleEpisodeMetrics(N, A, cash)
    _replay(ora, em, [tuple(b[key][t] for key in ACTION_KEYS) for t in range(T)])
    dev = [t.cpu().numpy() for t in env.collect_episode_metrics(module_of=module_of, n_modules=n_mod)]
    ref = em.table(module_of.cpu().numpy(), n_mod)
    assert (ref[0][:, 0] > 0).sum() >= k + 3                   # the trainable policies, random modules and both champions all played
    assert_tables_equal(dev[0], dev[1], *ref, what="league")
    from gym_continuousdoubleauction_amd import episode_metrics as EM
    s = EM.summarise(torch.from_numpy(dev[0]), torch.from_numpy(dev[1]), module_names=mapper.available_modules)
    assert set(s["modules"]) <= set(mapper.available_modules) and "policy_0" in s["modules"] and s["episodes"] == 2 * N
    env.close(); ora.close()


@pytest.mark.parametrize("path", ["step", "step_info", "rollout"])
def test_a_ledger_fault_in_an_episode_that_auto_resets_mid_run_is_reported(path):
    """a seeded corruption of ONE market's ledger (cash created out of nothing) in an episode that ends and resets itself in the middle of the run: the violation is
    counted, the market carries the sticky flag after its reset, the trainer-side check raises like the reference's strict_nav_check run"""
    from decimal import Decimal
    from gym_continuousdoubleauction_amd import CDAVecEnv, _capi as K, mlp
    from gym_continuousdoubleauction_amd import episode_metrics as EM
    N, A, max_step, victim = 64, 4, 6, 37
    env = CDAVecEnv(_cfg(A, max_step), n_markets=N, with_info=(path == "step_info"))
    env.reset(seed=99)
    env.enable_episode_metrics(True)
    rng = np.random.default_rng(2)
    for _ in range(2):
        env.step(*_actions(rng, N, A))
    st = env.get_state(victim)
    for field in ("cash", "nav", "prev_nav", "max_nav"):      # agent 2 finds 1234.5 that nobody lost
        d = getattr(st.acc[2], field)
        setattr(st.acc[2], field, K.decimal_to_dec(K.dec_to_decimal(d) + Decimal("1234.5")))
    env.set_state(victim, st)
    if path == "rollout":
        roll = mlp.RolloutChains(env, mlp.FusedPolicy("cuda:0", seed=3), 16, groups=2, seed=1)
        roll.run()
    else:
        for _ in range(16):
            env.step(*_actions(rng, N, A))
    torch.cuda.synchronize()
    flags = env.flags().cpu().numpy()
    assert flags[victim] & K.FLAG_NAV_CONSERVATION and not (np.delete(flags, victim) & K.FLAG_NAV_CONSERVATION).any()
    agent, envrow = env.collect_episode_metrics()
    s = EM.summarise(agent, envrow)
    assert s["nav_conservation_violations"] == 1 and s["nav_conservation_error"] == 1234.5 and s["episodes"] >= 2 * N
    with pytest.raises(EM.NavConservationError):
        EM.check_nav_conservation(0, s, strict=True)
    logged = []
    EM.check_nav_conservation(0, s, strict=False, log=logged.append)
    assert logged and "strict_nav_check is off" in logged[0]
    _, bad = env.nav_conservation()                            # ... while the end-of-run check sees nothing: the faulty episode is long gone
    assert not bool(bad.any())
    env.close()


def test_metrics_off_touch_nothing_and_the_step_is_unchanged():
    """off (the default): no tallies, no accumulators, the same outputs bit for bit as with metrics on"""
    from gym_continuousdoubleauction_amd import CDAVecEnv
    N, A = 64, 4
    outs = []
    for on in (False, True):
        env = CDAVecEnv(_cfg(A, 5), n_markets=N, with_info=False)
        env.reset(seed=3)
        if on:
            env.enable_episode_metrics(True)
        rng = np.random.default_rng(8)
        rows = []
        for _ in range(12):
            o, r, te, tr, _ = env.step(*_actions(rng, N, A))
            rows.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), te.cpu().numpy().copy(), tr.cpu().numpy().copy()))
        agent, envrow = env.collect_episode_metrics()
        assert bool(envrow[0] > 0) == on and bool(agent.any()) == on
        outs.append((rows, bytes(env.get_state(5))))
        env.close()
    for (a, b) in zip(outs[0][0], outs[1][0]):
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    assert outs[0][1] == outs[1][1]
