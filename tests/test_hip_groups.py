"""GPU: the batch stepped as concurrent market groups (cda_step_groups / cda_step_range) gives bit-identical results
to one launch over all markets and to the CPU oracle; the device-side random-agent stream equals the host one; the
book census equals the oracle's; a never-reset env steps on defined state."""
import numpy as np
import pytest

from gym_continuousdoubleauction_amd import _capi as K

pytestmark = pytest.mark.gpu


def _np_info(env):
    import torch  # noqa: F401
    from gym_continuousdoubleauction_amd.vec_env import DEC_DTYPE
    out = {}
    for k, v in env.info.items():
        a = v.cpu().numpy()
        out[k] = a.view(DEC_DTYPE).reshape(env.n_markets, env.num_agents) if k == "nav" else a
    return out


@pytest.mark.parametrize("groups", [2, 3, 5])
def test_groups_equal_single_launch_and_oracle(groups):
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    import oracle_lib as O
    n, a, steps = 1000, 4, 48                        # 1000 is not divisible by 3: uneven groups
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": steps, "is_render": False}
    one, grp, ora = CDAVecEnv(cfg, n), CDAVecEnv(cfg, n, groups=groups), O.OracleEnv(cfg, n)
    assert sum(c for _, c in grp.group_ranges) == n and grp.group_ranges[0][0] == 0
    seeds = np.arange(31000, 31000 + n, dtype=np.uint64)
    o1 = one.reset(seed=seeds).cpu().numpy()
    o2 = grp.reset(seed=seeds).cpu().numpy()
    assert np.array_equal(o1.view(np.uint32), o2.view(np.uint32))
    ora.reset(seeds)
    for t in range(steps):
        acts = one.random_actions(t, action_seed=99, market_index_base=5)
        r1 = one.step(*acts)
        r2 = grp.step(*acts)
        grp.join()                                   # outputs are consumed on the current stream below
        oo, orw, ot, otr, oi = ora.step(*acts)
        for x, y in zip(r1[:4], r2[:4]):
            assert torch.equal(x, y), t
        i1, i2 = _np_info(one), _np_info(grp)
        for k in i1:
            assert np.array_equal(np.ascontiguousarray(i1[k]).view(np.uint8), np.ascontiguousarray(i2[k]).view(np.uint8)), (k, t)
            assert np.array_equal(np.ascontiguousarray(i2[k]).view(np.uint8), np.ascontiguousarray(oi[k]).view(np.uint8)), (k, t)
        assert np.array_equal(r2[0].cpu().numpy().view(np.uint32), oo.view(np.uint32)), t
        assert np.array_equal(r2[1].cpu().numpy().view(np.uint64), orw.view(np.uint64)), t
    for i in list(range(0, n, 37)) + [n - 1]:
        assert bytes(grp.get_state(i)) == bytes(one.get_state(i)) == bytes(ora.get_state(i)), i
    assert np.array_equal(grp.book_peak().cpu().numpy(), ora.book_peak())
    assert (grp.flags() == 0).all()
    for e in (one, grp, ora):
        e.close()


def test_step_range_leaves_other_markets_untouched():
    import ctypes as C
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd._lib import lib, check
    n, a = 64, 4
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 64, "is_render": False}
    env = CDAVecEnv(cfg, n, with_info=False)
    env.reset(seed=7)
    before = [bytes(env.get_state(i)) for i in range(n)]
    cat, mean, sigma, price, off = env.random_actions_device(0, 1, action_seed=3)
    st = torch.cuda.current_stream().cuda_stream
    check(lib().cda_step_range(env._h, 16, 24, cat.data_ptr(), mean.data_ptr(), sigma.data_ptr(), price.data_ptr(), off.data_ptr(), None,
                               env.obs.data_ptr(), env.reward.data_ptr(), env._term.data_ptr(), env._trunc.data_ptr(), None, st), "cda_step_range")
    after = [bytes(env.get_state(i)) for i in range(n)]
    for i in range(n):
        assert (after[i] != before[i]) == (16 <= i < 40), i
    # out-of-range requests are refused
    assert lib().cda_step_range(env._h, 60, 8, cat.data_ptr(), mean.data_ptr(), sigma.data_ptr(), price.data_ptr(), off.data_ptr(), None,
                                env.obs.data_ptr(), env.reward.data_ptr(), env._term.data_ptr(), env._trunc.data_ptr(), None, st) == K.ERR_INVALID
    env.close()


def test_device_random_actions_equal_the_host_sampler():
    from gym_continuousdoubleauction_amd import CDAVecEnv
    n, a = 300, 8
    env = CDAVecEnv({"num_of_agents": a, "is_render": False}, n, with_info=False)
    dev = env.random_actions_device(5, 7, action_seed=2024, market_index_base=4096)
    for s in range(7):
        host = env.random_actions(5 + s, action_seed=2024, market_index_base=4096)
        for d, h in zip(dev, host):
            assert np.array_equal(d[s].cpu().numpy().view(np.uint32), h.view(np.uint32)), s
    env.close()


def test_never_reset_env_steps_on_defined_state():
    """cda_create builds every market the way the reference's __init__ does (accounts at init_cash, empty book): a step
    before the first reset is deterministic and raises no domain flag."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    cfg = {"num_of_agents": 4, "init_cash": 500000, "max_step": 8, "is_render": False}
    outs = []
    for _ in range(2):
        env = CDAVecEnv(cfg, 33)
        s = env.get_state(0)
        assert K.dec_to_decimal(s.acc[3].cash) == 500000 and K.dec_to_decimal(s.acc[0].nav) == 500000 and s.n_bids == s.n_asks == 0
        for t in range(4):
            obs, rew, term, trunc, info = env.step(*env.random_actions(t, action_seed=1))
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert (env.flags() == 0).all()
        outs.append((obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), bytes(env.get_state(32))))
        env.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


def test_config_scale_beyond_int32_level_sums_is_refused():
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd._lib import CDAError
    with pytest.raises(CDAError):
        CDAVecEnv({"num_of_agents": 2, "mkt_max_size": 1 << 20, "limit_size_multiple": 64, "is_render": False}, 1)
    CDAVecEnv({"num_of_agents": 2, "mkt_max_size": 100000, "limit_size_multiple": 20, "is_render": False}, 1).close()


def test_book_capacity_variants_512_pool_for_many_agents():
    """Two compiled pools (cda_config.book_capacity): 16 agents get 512 resting orders per market by default - the flip-heavy law
    overflows 256 within a few thousand steps (profiles/r02 census) - and the larger build is bit-identical to the oracle and, while
    nothing overflows, to the 256 build."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    import oracle_lib as O
    n, a, steps = 96, 16, 160
    cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": 4096, "is_render": False}
    big, small, ora = CDAVecEnv(cfg, n), CDAVecEnv(dict(cfg, book_capacity=256), n), O.OracleEnv(cfg, n)
    assert big.book_capacity == 512 and small.book_capacity == 256
    seeds = np.arange(77, 77 + n, dtype=np.uint64)
    big.reset(seed=seeds); small.reset(seed=seeds); ora.reset(seeds)
    rng = np.random.default_rng(4)
    for t in range(steps):
        # the flip-heavy law of tests/golden/make_goldens.py
        acts = (rng.choice([1, 2, 2, 5, 6, 6, 3, 7, 4, 8], (n, a)).astype(np.int32), rng.uniform(-0.05, 0.05, (n, a)).astype(np.float32),
                rng.uniform(0, 1, (n, a)).astype(np.float32), rng.integers(0, 3, (n, a)).astype(np.int32), rng.choice([1, 2, 2], (n, a)).astype(np.int32))
        ob, rb, *_ = big.step(*acts)
        small.step(*acts)
        oo, orw, *_ = ora.step(*acts)
        assert np.array_equal(ob.cpu().numpy().view(np.uint32), oo.view(np.uint32)) and np.array_equal(rb.cpu().numpy().view(np.uint64), orw.view(np.uint64)), t
    assert (big.flags() == 0).all() and (big.check_invariants() == 0).all()
    assert np.array_equal(big.book_peak().cpu().numpy(), ora.book_peak())
    clear = (small.flags() == 0).cpu().numpy()
    for i in range(n):
        assert bytes(big.get_state(i)) == bytes(ora.get_state(i)), i
        if clear[i]:
            assert bytes(small.get_state(i)) == bytes(big.get_state(i)), i
    # a 4-agent env on the 512 pool equals the default 256 one
    c4 = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 64, "is_render": False}
    x, y = CDAVecEnv(c4, 64), CDAVecEnv(dict(c4, book_capacity=512), 64)
    x.reset(seed=5); y.reset(seed=5)
    for t in range(48):
        acts = x.random_actions(t, action_seed=8)
        rx, ry = x.step(*acts), y.step(*acts)
        assert torch.equal(rx[0], ry[0]) and torch.equal(rx[1], ry[1])
    assert all(bytes(x.get_state(i)) == bytes(y.get_state(i)) for i in range(0, 64, 7))
    for e in (big, small, ora, x, y):
        e.close()
