"""GPU, BASELINE config #2 (1024 parallel markets x 4 random agents, bit-exact LOB check vs CPU) and the
A=8 variant: the HIP path vs the CPU oracle on identical seeded action streams, every output tensor compared
bit for bit at every step and the full per-market state dump at the end; plus size-independent properties at
the full roofline size (4096 markets)."""
from decimal import Decimal

import numpy as np
import pytest

from gym_continuousdoubleauction_amd import _capi as K

pytestmark = pytest.mark.gpu


def _actions(rng, n, a):
    return (rng.integers(0, 9, (n, a)).astype(np.int32), rng.uniform(-1, 1, (n, a)).astype(np.float32),
            rng.uniform(0, 1, (n, a)).astype(np.float32), rng.integers(0, 10, (n, a)).astype(np.int32),
            rng.integers(0, 3, (n, a)).astype(np.int32))


# 6000 markets exceed what is resident at 4 waves per SIMD: workgroups of that case run in two rounds
@pytest.mark.parametrize("n,a,steps", [(1024, 4, 256), (512, 8, 128), (64, 16, 64), (37, 5, 96), (6000, 4, 48)])
def test_hip_equals_oracle_every_step(n, a, steps):
    from hip_env import HipEnv
    import oracle_lib as O
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": steps, "is_render": False}
    env, ora = HipEnv(cfg, n), O.OracleEnv(cfg, n)
    seeds = np.arange(1000, 1000 + n, dtype=np.uint64)          # market i seeded SeedSequence(1000 + i)
    assert np.array_equal(env.reset(seeds).view(np.uint32), ora.reset(seeds).view(np.uint32))
    rng = np.random.default_rng(2024)
    for t in range(steps):
        acts = _actions(rng, n, a)
        assert np.array_equal(env.raw_snapshot().view(np.uint32), ora.raw_snapshot().view(np.uint32)), t
        obs, rew, term, trunc, info = env.step(*acts)
        oo, orw, ot, otr, oi = ora.step(*acts)
        assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32)), f"obs, step {t}"
        assert np.array_equal(rew.view(np.uint64), orw.view(np.uint64)), f"reward, step {t}"
        assert np.array_equal(term, ot) and np.array_equal(trunc, otr)
        for k in oi:
            x, y = info[k], oi[k]
            assert np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)), f"info.{k}, step {t}"
    assert trunc.all()
    for i in list(range(0, n, max(1, n // 64))) + [n - 1]:
        assert bytes(env.get_state(i)) == bytes(ora.get_state(i)), f"state of market {i}"
    assert (env.flags() == 0).all()
    env.close(); ora.close()


def test_properties_at_roofline_size():
    """4096 markets x 4 agents (BASELINE config #3): invariants the reference's own harness checks."""
    from hip_env import HipEnv
    n, a, steps = 4096, 4, 200
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 100000, "is_render": False}
    env = HipEnv(cfg, n)
    env.reset(np.arange(5000, 5000 + n, dtype=np.uint64))
    rng = np.random.default_rng(7)
    for t in range(steps):
        obs, rew, term, trunc, info = env.step(*_actions(rng, n, a))
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    assert not term.any() and not trunc.any()
    # every unit long is a unit short (exact); NAV is conserved (league callback :679-704) up to the
    # 28-digit rounding noise of the ledger
    assert (info["net_position"].sum(axis=1) == 0).all()
    nav = env.env.nav_decimals()
    worst = max(abs(sum(row) - Decimal(a) * 1000000) for row in nav)
    assert worst < Decimal("1e-15"), worst
    assert (env.flags() == 0).all()
    for i in range(0, n, 97):
        s = env.get_state(i)
        bids = [s.bids[j] for j in range(s.n_bids)]
        asks = [s.asks[j] for j in range(s.n_asks)]
        assert all(bids[j].price >= bids[j + 1].price for j in range(len(bids) - 1))       # sorted
        assert all(asks[j].price <= asks[j + 1].price for j in range(len(asks) - 1))
        assert not (bids and asks) or bids[0].price < asks[0].price                         # never crossed
        assert all(o.qty > 0 for o in bids + asks)
        # escrow equals the value of the resting orders, exactly (cash_on_hold stays integer valued)
        for tr in range(a):
            held = sum(o.price * o.qty for o in bids + asks if o.owner == tr)
            assert K.dec_to_decimal(s.acc[tr].cash_on_hold) == Decimal(held)
    env.close()


def test_market_results_do_not_depend_on_the_batch():
    """Market i steps identically whether it runs alone, in a batch of 8 or of 300 (independence + determinism)."""
    from hip_env import HipEnv
    a, steps = 4, 64
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 100000, "is_render": False}
    rng = np.random.default_rng(11)
    acts = [_actions(rng, 300, a) for _ in range(steps)]
    outs = []
    for n in (1, 8, 300, 300):
        env = HipEnv(cfg, n)
        env.reset(np.arange(77, 77 + n, dtype=np.uint64))
        rec = []
        for t in range(steps):
            obs, rew, _, _, _ = env.step(*[x[:n] for x in acts[t]])
            rec.append((obs[0].copy(), rew[0].copy()))
        outs.append((rec, bytes(env.get_state(0))))
        env.close()
    for rec, st in outs[1:]:
        assert st == outs[0][1]
        for (o1, r1), (o2, r2) in zip(rec, outs[0][0]):
            assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32)) and np.array_equal(r1.view(np.uint64), r2.view(np.uint64))


def test_book_capacity_overflow_is_flagged_not_silent():
    """An env built WITHOUT the HBM tier (book_spill = -1: the tile is the whole book): more than CDA_BOOK_CAP resting orders -
    the extra rest is dropped and the market flagged."""
    from hip_env import HipEnv
    import oracle_lib as O
    cfg = {"num_of_agents": 2, "init_cash": 10 ** 9, "max_step": 64, "is_render": False, "book_spill": -1}
    env, ora = HipEnv(cfg, 1), O.OracleEnv(cfg, 1)
    for e in (env, ora):
        e.reset(np.array([3], np.uint64))
        s = e.get_state(0)
        s.n_bids = K.BOOK_CAP
        for i in range(K.BOOK_CAP):
            o = s.bids[i]
            o.price, o.qty, o.owner, o.order_id, o.timestamp = 10000 - i, 1, 0, i + 1, i + 1
        s.lob_time = s.next_order_id = K.BOOK_CAP
        e.set_state(0, s)
        e.place_order(0, 1, K.T_LIMIT, K.S_BID, 5, 50)
    assert env.flags()[0] & K.FLAG_BOOK_OVERFLOW and ora.flags()[0] & K.FLAG_BOOK_OVERFLOW
    assert bytes(env.get_state(0)) == bytes(ora.get_state(0))
    assert env.get_state(0).n_bids == K.BOOK_CAP
    env.close(); ora.close()


@pytest.mark.parametrize("with_info", [True, False])     # with info tensors: a k_reset launch behind the step; without: the step kernel's own last act
def test_auto_reset_equals_step_then_masked_reset(with_info):
    """cda_config.auto_reset: a market whose episode ended is reset on the device right after the step (seed=None
    semantics).  Equivalent to the oracle's step followed by reset(mask=terminated|truncated): same rewards and
    flags for the finished step, the new episode's first observation in the obs row, same states afterwards."""
    from hip_env import HipEnv
    import oracle_lib as O
    n, a, steps = 96, 4, 70
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 16, "is_render": False}
    hip = HipEnv(dict(cfg, auto_reset=True), n, with_info=with_info)
    orc = O.OracleEnv(cfg, n_markets=n)
    seeds = np.arange(900, 900 + n, dtype=np.uint64)
    assert np.array_equal(hip.reset(seeds), orc.reset(seeds))
    rng = np.random.default_rng(5)
    n_resets = 0
    for t in range(steps):
        acts = _actions(rng, n, a)
        if t == 20:                                       # desynchronise the episodes: reseed a third of the markets mid-run
            m = (np.arange(n) % 3 == 0).astype(np.uint8)
            assert np.array_equal(hip.reset(None, m)[m == 1], orc.reset(None, m)[m == 1])
        ho, hr, ht, hu, _ = hip.step(*acts)
        oo, orw, ot, ou, _ = orc.step(*acts)
        oo, orw, ot, ou = oo.copy(), orw.copy(), ot.copy(), ou.copy()
        done = ((ot != 0) | (ou != 0)).astype(np.uint8)
        if done.any():
            oo[done == 1] = orc.reset(None, done)[done == 1]
            n_resets += int(done.sum())
        assert np.array_equal(hr.view(np.uint64), orw.view(np.uint64)), t
        assert np.array_equal(ht, ot) and np.array_equal(hu, ou), t
        assert np.array_equal(ho.view(np.uint32), oo.view(np.uint32)), t
    assert n_resets >= 3 * n
    for i in range(0, n, 7):
        assert bytes(hip.get_state(i)) == bytes(orc.get_state(i)), f"state of market {i}"
    hip.close(); orc.close()


@pytest.mark.parametrize("with_info", [True, False])
def test_auto_reset_of_markets_whose_books_outgrew_the_tile(with_info):
    """The same equivalence where the general build steps the market (books of 2 x 400 resting orders: the HBM tier): the episode's end resets
    a deep book too - inside slow_step's own tail in the info-less kernel."""
    from hip_env import HipEnv
    from fuzz_cases import prefill_book
    import oracle_lib as O
    n, a = 12, 4
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 5, "is_render": False}
    hip = HipEnv(dict(cfg, auto_reset=True), n, with_info=with_info)
    orc = O.OracleEnv(cfg, n_markets=n)
    seeds = np.arange(40, 40 + n, dtype=np.uint64)
    assert np.array_equal(hip.reset(seeds), orc.reset(seeds))
    rng = np.random.default_rng(9)
    n_resets = 0
    for t in range(12):
        if t in (0, 6):                                   # deep books in four markets at the start of two episodes
            for i in range(0, n, 3):
                for e in (hip, orc):
                    prefill_book(e, i, np.random.default_rng(70 + i + t), a, 400, 400)
        acts = (rng.integers(0, 9, (n, a)).astype(np.int32), rng.uniform(-0.01, 0.01, (n, a)).astype(np.float32), rng.uniform(0, 1, (n, a)).astype(np.float32),
                rng.integers(0, 10, (n, a)).astype(np.int32), rng.integers(0, 3, (n, a)).astype(np.int32))
        ho, hr, ht, hu, _ = hip.step(*acts)
        oo, orw, ot, ou, _ = orc.step(*acts)
        oo, orw, ot, ou = oo.copy(), orw.copy(), ot.copy(), ou.copy()
        done = ((ot != 0) | (ou != 0)).astype(np.uint8)
        if done.any():
            oo[done == 1] = orc.reset(None, done)[done == 1]
            n_resets += int(done.sum())
        assert np.array_equal(hr.view(np.uint64), orw.view(np.uint64)), t
        assert np.array_equal(ht, ot) and np.array_equal(hu, ou), t
        assert np.array_equal(ho.view(np.uint32), oo.view(np.uint32)), t
    assert n_resets >= 2 * n
    for i in range(n):
        assert bytes(hip.get_state(i)) == bytes(orc.get_state(i)), f"state of market {i}"
        for side in (0, 1):
            assert np.array_equal(hip.get_book(i, side), orc.get_book(i, side))
    assert int((hip.flags() != 0).sum()) == 0
    hip.close(); orc.close()


def test_fused_random_agent_episode_equals_stepwise_and_oracle():
    """cda_run_random: a whole random-agent episode per launch (state in LDS from step to step, every market at its own
    pace) must be bit-identical to stepping the same markets one launch per step on the same actions, and to the oracle."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    import oracle_lib as O
    n, a, horizon, seed = 192, 4, 48, 20240927
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": horizon, "is_render": False}
    fused, stepw, chunk = (CDAVecEnv(cfg, n_markets=n, with_info=False) for _ in range(3))
    orc = O.OracleEnv(cfg, n_markets=n)
    seeds = np.arange(4000, 4000 + n, dtype=np.uint64)
    for e in (fused, stepw, chunk):
        e.reset(seed=seeds)
    orc.reset(seeds)
    # bankrupt a few accounts of market 5 so that at least one market ends early by termination, not truncation
    obs, ret, term, trunc, steps = fused.run_random(64, action_seed=seed, market_index_base=7)
    obs, ret, term, trunc, steps = obs.cpu().numpy(), ret.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy(), steps.cpu().numpy()
    assert (steps == horizon).all() and trunc.all() and not term.any()
    ret_ref = np.zeros((n, a), np.float64)
    for t in range(horizon):
        acts = stepw.random_actions(t, action_seed=seed, market_index_base=7)
        so, sr, st, su, _ = stepw.step(*acts)
        oo, orw, ot, ou, _ = orc.step(*acts)
        assert np.array_equal(sr.cpu().numpy().view(np.uint64), orw.view(np.uint64)), t
        ret_ref += orw                                          # the same left-to-right f64 sum the kernel forms
    assert np.array_equal(obs.view(np.uint32), so.cpu().numpy().view(np.uint32))
    assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32))
    assert np.array_equal(ret.view(np.uint64), ret_ref.view(np.uint64))
    for i in list(range(0, n, 13)) + [n - 1]:
        assert bytes(fused.get_state(i)) == bytes(stepw.get_state(i)) == bytes(orc.get_state(i)), f"state of market {i}"
    # the sampler is keyed by the market's own step counter: 20 + 28 steps in two launches are the same episode
    # (2, 1, 0 and 17 steps first: fewer steps than history frames leaves older frames of the ring untouched)
    for part in (2, 1, 0, 17):
        chunk.run_random(part, action_seed=seed, market_index_base=7)
        assert (chunk._rr_steps.cpu().numpy() == part).all()
    o2, r2, t2, u2, s2 = chunk.run_random(64, action_seed=seed, market_index_base=7)
    assert (s2.cpu().numpy() == horizon - 20).all() and u2.cpu().numpy().all()
    assert np.array_equal(o2.cpu().numpy().view(np.uint32), obs.view(np.uint32))
    for i in range(0, n, 29):
        assert bytes(chunk.get_state(i)) == bytes(fused.get_state(i))
    # the action law: ranges, and roughly uniform categories
    cat, mean, sigma, price, off = fused.random_actions(3, action_seed=1)
    assert cat.min() >= 0 and cat.max() <= 8 and price.min() >= 0 and price.max() <= 9 and off.min() >= 0 and off.max() <= 2
    assert mean.min() >= -1 and mean.max() < 1 and sigma.min() >= 0 and sigma.max() < 1
    big = np.concatenate([fused.random_actions(t, action_seed=9)[0].ravel() for t in range(40)])
    assert np.abs(np.bincount(big, minlength=9) / big.size - 1 / 9).max() < 0.01
    assert (fused.flags() == 0).all()
    for e in (fused, stepw, chunk):
        e.close()
    orc.close()


def test_device_nav_conservation_check_equals_the_reference_rule():
    """cda_nav_conservation vs the reference's own statement of the invariant (callbk:679-704) in CPython Decimal."""
    from hip_env import HipEnv
    n, a, cash = 256, 4, 1000000
    env = HipEnv({"num_of_agents": a, "init_cash": cash, "max_step": 1000, "is_render": False}, n)
    env.reset(np.arange(77, 77 + n, dtype=np.uint64))
    rng = np.random.default_rng(11)
    for t in range(120):
        env.step(*_actions(rng, n, a))
    err, bad = env.env.nav_conservation(1e-6)
    err, bad = err.cpu().numpy(), bad.cpu().numpy()
    for i in range(n):
        st = env.get_state(i)
        total = Decimal(0)
        for k in range(a):
            total += K.dec_to_decimal(st.acc[k].nav)
        want = float(abs(total - Decimal(str(cash)) * a))
        assert err[i] == want, (i, err[i], want)
    assert not bad.any() and err.max() < 1e-15
    s = env.get_state(5)                                       # break the invariant of one market
    s.acc[2].nav.w[2] += 1                                    # + 2^64 units of the coefficient: ~0.02 at exponent -21
    env.set_state(5, s)
    err2, bad2 = env.env.nav_conservation(1e-6)
    assert bool(bad2[5]) and int(bad2.sum()) == 1 and float(err2[5]) > 0
    env.close()


def test_hip_equals_oracle_on_random_configurations():
    """Fuzz over the config space (agent counts up to the bound, balances from 400 to 5e10, size and price ranges, history
    depths, reward coefficients, action laws, agent subsets): every output and the final states, HIP vs oracle.  The same
    generator drives tests/golden/crosscheck_oracle.py, which pins the oracle to the reference on such episodes."""
    from hip_env import HipEnv
    import oracle_lib as O
    from fuzz_cases import batch_actions, prefill_book, random_config, random_order
    import os
    rng = np.random.default_rng(int(os.environ.get("CDA_FUZZ_SEED", "31337")))
    for case in range(int(os.environ.get("CDA_FUZZ_CASES", "18"))):       # a longer soak: CDA_FUZZ_CASES=150 CDA_FUZZ_SEED=...
        cfg, law, present_p = random_config(rng)
        order = random_order(rng)
        n, a, steps = 40, cfg["num_of_agents"], 56
        env, ora = HipEnv(cfg, n), O.OracleEnv(cfg, n)
        seeds = rng.integers(0, 2 ** 63, n).astype(np.uint64)
        assert np.array_equal(env.reset(seeds).view(np.uint32), ora.reset(seeds).view(np.uint32)), (case, cfg)
        if rng.random() < 0.4:            # deep books from the first step on (far beyond the LDS tile: the HBM tier is in play at once)
            pseed = int(rng.integers(0, 2 ** 31))
            for i in range(0, n, 3):
                nb, na = (int(x) for x in np.random.default_rng(pseed + i).integers(0, 513, 2))
                for e in (env, ora):
                    prefill_book(e, i, np.random.default_rng(pseed + i + 1), a, nb, na)
        for t in range(steps):
            acts, present = batch_actions(rng, n, a, law, present_p, order)
            obs, rew, term, trunc, info = env.step(*acts, present)
            oo, orw, ot, otr, oi = ora.step(*acts, present)
            ctx = f"case {case} {cfg} law={law} step {t}"
            assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32)), ctx
            assert np.array_equal(rew.view(np.uint64), orw.view(np.uint64)), ctx
            assert np.array_equal(term, ot) and np.array_equal(trunc, otr), ctx
            for k in oi:
                assert np.array_equal(np.ascontiguousarray(info[k]).view(np.uint8), np.ascontiguousarray(oi[k]).view(np.uint8)), f"{ctx} info.{k}"
        for i in range(0, n, 3):
            assert bytes(env.get_state(i)) == bytes(ora.get_state(i)), f"case {case} market {i}"
            for side in (0, 1):
                assert np.array_equal(env.get_book(i, side), ora.get_book(i, side)), f"case {case} market {i} side {side}"
        assert np.array_equal(env.flags(), ora.flags()), case
        assert (env.env.check_invariants().cpu().numpy() == 0).all(), case
        env.close(); ora.close()


def test_fused_episodes_equal_stepwise_on_random_configurations():
    """cda_run_random vs one launch per step over random configurations (agent counts, balances, sizes, history depths)."""
    import os
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from fuzz_cases import random_config
    rng = np.random.default_rng(int(os.environ.get("CDA_FUZZ_SEED", "777")))
    for case in range(int(os.environ.get("CDA_FUZZ_CASES", "10"))):
        cfg, _, _ = random_config(rng)
        cfg["max_step"] = int(rng.integers(5, 40))
        n, T = 48, cfg["max_step"] + 3
        fused, stepw = CDAVecEnv(cfg, n_markets=n, with_info=False), CDAVecEnv(cfg, n_markets=n, with_info=False)
        seeds = rng.integers(0, 2 ** 63, n).astype(np.uint64)
        fused.reset(seed=seeds); stepw.reset(seed=seeds)
        aseed, base = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 10 ** 6))
        obs, ret, term, trunc, steps = fused.run_random(T, action_seed=aseed, market_index_base=base)
        ret_ref = np.zeros((n, cfg["num_of_agents"]))
        alive = np.ones(n, bool)
        taken = np.zeros(n, np.int32)
        for t in range(cfg["max_step"]):                       # every market runs to truncation unless it terminates first
            so, sr, st, su, _ = stepw.step(*stepw.random_actions(t, action_seed=aseed, market_index_base=base))
            sr, st, su = sr.cpu().numpy(), st.cpu().numpy(), su.cpu().numpy()
            ret_ref[alive] += sr[alive]
            taken[alive] += 1
            alive &= ~(st | su)
            if not alive.any():
                break
        assert np.array_equal(steps.cpu().numpy(), taken), (case, cfg)
        assert np.array_equal(ret.cpu().numpy().view(np.uint64), ret_ref.view(np.uint64)), (case, cfg)
        full = taken == cfg["max_step"]                        # markets that ended together with the stepwise run: same final state
        assert full.any()
        for i in np.flatnonzero(full)[::7]:
            assert bytes(fused.get_state(int(i))) == bytes(stepw.get_state(int(i))), (case, int(i))
        assert np.array_equal(obs.cpu().numpy()[full].view(np.uint32), so.cpu().numpy()[full].view(np.uint32)), case
        fused.close(); stepw.close()
