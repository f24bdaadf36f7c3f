"""GPU: the batched attach point `CDAVecMultiAgentEnv` (N markets behind one object) and the facade fixes of round 2:
one device-to-host copy per step, writable account diagnostics, entropy-seeded first reset."""
import json
import time
from decimal import Decimal

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 200, "is_render": False}


def _rand_dict(rng, agents):
    return {a: {"category": np.int64(rng.integers(0, 9)), "size_mean": rng.uniform(-1, 1, 1).astype(np.float32),
                "size_sigma": rng.uniform(0, 1, 1).astype(np.float32), "price": np.int64(rng.integers(0, 10)),
                "price_offset": np.int64(rng.integers(0, 3))} for a in agents}


def _bits_equal(x, y):
    """bitwise equality (NaN encodes None in best_bid / best_ask / spread)"""
    import torch
    return x.shape == y.shape and torch.equal(x.contiguous().view(torch.uint8), y.contiguous().view(torch.uint8))


def test_step_batch_returns_views_that_equal_the_vec_env_outputs():
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv, CDAVecMultiAgentEnv
    n, a = 96, 4
    me, ve = CDAVecMultiAgentEnv(CFG, num_envs=n), CDAVecEnv(CFG, n_markets=n)
    obs_m, infos = me.reset_batch(seed=40)
    obs_v = ve.reset(seed=40)
    assert set(obs_m) == set(me.agents) and all(infos[x] == {} for x in me.agents)
    assert all(obs_m[x] is obs_m["agent_0"] for x in me.agents) and torch.equal(obs_m["agent_0"], obs_v)
    for t in range(24):
        acts = ve.random_actions_device(t, 1, action_seed=9)
        batch = {k: v[0] for k, v in zip(("category", "size_mean", "size_sigma", "price", "price_offset"), acts)}
        if t % 2:        # the per-agent spelling of the same actions: {agent: {key: [N]}}
            batch = {x: {k: batch[k][:, j] for k in batch} for j, x in enumerate(me.agents)}
        obs, rew, term, trunc, info = me.step_batch(batch)
        vo, vr, vt, vtr, vi = ve.step(*[v[0] for v in acts])
        base = me.vec.packed.data_ptr()
        for j, x in enumerate(me.agents):
            assert obs[x].data_ptr() == me.vec.obs.data_ptr() and torch.equal(obs[x], vo)            # a view, not a copy
            assert rew[x].data_ptr() == me.vec.reward.data_ptr() + 8 * j and torch.equal(rew[x], vr[:, j])
            assert not term[x].any() and not trunc[x].any()
            for name, tns in info[x].items():
                assert base <= tns.data_ptr() < base + me.vec.packed.numel(), name                   # inside the env's own buffer
                ref = vi[name]
                assert _bits_equal(tns, ref[:, j] if ref.dim() >= 2 and ref.shape[1] == a else ref), (name, t)
        assert torch.equal(term["__all__"], vt) and torch.equal(trunc["__all__"], vtr)
    me.close(); ve.close()


def test_vector_protocol_equals_one_facade_per_market():
    """reset()/step() over lists of per-market dicts: market i's dicts are exactly what a CDAEnv seeded seed + i returns."""
    from gym_continuousdoubleauction_amd import CDAEnv, CDAVecMultiAgentEnv
    n = 5
    me = CDAVecMultiAgentEnv(CFG, num_envs=n)
    singles = [CDAEnv(CFG) for _ in range(n)]
    obs_l, info_l = me.reset(seed=700)
    for i, e in enumerate(singles):
        o, inf = e.reset(seed=700 + i)
        assert np.array_equal(obs_l[i]["agent_2"], o["agent_2"]) and info_l[i] == inf
    rng = np.random.default_rng(3)
    for t in range(30):
        dicts = [_rand_dict(rng, me.agents) for _ in range(n)]
        if t % 5 == 0:
            del dicts[1]["agent_2"]                               # a subset of the agents acts
        outs = me.step(dicts)
        for i, e in enumerate(singles):
            o, r, te, tr, inf = e.step(dicts[i])
            assert np.array_equal(outs[0][i]["agent_0"], o["agent_0"]) and outs[0][i]["agent_0"] is outs[0][i]["agent_3"]
            assert outs[1][i] == r and outs[2][i] == te and outs[3][i] == tr
            assert json.dumps(outs[4][i], sort_keys=True) == json.dumps(inf, sort_keys=True)
    me.close()
    for e in singles:
        e.close()


def test_account_fields_are_writable_like_the_reference_tests_do():        # test/test_accounting.py:143-150
    from gym_continuousdoubleauction_amd import CDAEnv
    env = CDAEnv(CFG)
    obs0, _ = env.reset(seed=9)
    acc = env.traders[1].acc
    acc.cash = Decimal(900)
    acc.net_position = -7
    acc.VWAP = Decimal("12.5")
    acc.position_val = Decimal("87.5")
    assert acc.cash == Decimal(900) and acc.net_position == -7 and acc.VWAP == Decimal("12.5") and acc.position_val == Decimal("87.5")
    assert env.traders[0].acc.cash == Decimal(1000000)                       # the other accounts are untouched
    with pytest.raises(AttributeError):
        acc.no_such_field = 1
    with pytest.raises(ValueError):
        acc.net_position = 1.5
    # the write reached the device: the next step settles against it (a 900-cash account cannot afford a big bid)
    big_bid = {"category": np.int64(2), "size_mean": np.array([1.0], np.float32), "size_sigma": np.array([0.0], np.float32),
               "price": np.int64(0), "price_offset": np.int64(1)}
    _, _, _, _, infos = env.step({"agent_1": big_bid})
    assert infos["agent_1"]["num_rejected_step"] == 1 and infos["agent_1"]["cash"] == 900.0
    obs, *_ = env.step({})
    assert np.array_equal(obs["agent_0"][: 2 * 42], obs0["agent_0"][2 * 42:])     # the history ring survived the state write
    env.close()


def test_first_unseeded_reset_draws_entropy_later_ones_continue_the_stream():     # ADVICE r1 (gymnasium np_random semantics)
    from gym_continuousdoubleauction_amd import CDAEnv
    firsts = set()
    for _ in range(6):
        env = CDAEnv(dict(CFG, initial_price_min=1, initial_price_max=100000))
        env.reset()
        s1 = env._vec.get_state(0)
        firsts.add((s1.rng_state_hi, s1.rng_state_lo))
        inc = (s1.rng_inc_hi, s1.rng_inc_lo)
        env.reset()                                              # seed=None again: same generator, stream continues
        s2 = env._vec.get_state(0)
        # (numpy buffers half of every 64-bit draw: the position in the stream is the state plus that buffer flag)
        pos = lambda s: (s.rng_state_hi, s.rng_state_lo, s.rng_has_uint32)   # noqa: E731
        assert (s2.rng_inc_hi, s2.rng_inc_lo) == inc and pos(s2) != pos(s1)
        env.close()
    assert len(firsts) == 6                                      # six fresh envs, six different streams
    a, b = CDAEnv(CFG), CDAEnv(CFG)
    assert np.array_equal(a.reset(seed=5)[0]["agent_0"], b.reset(seed=5)[0]["agent_0"])       # seeded: reproducible
    a.close(); b.close()


def test_facade_steps_per_second_are_reported(capsys):
    """Not a threshold test: prints env-steps/s of the dict facades next to the reference's 2.26 k env-steps/s."""
    from gym_continuousdoubleauction_amd import CDAEnv, CDAVecMultiAgentEnv
    rng = np.random.default_rng(0)
    env = CDAEnv(CFG)
    env.reset(seed=1)
    acts = [_rand_dict(rng, env.agents) for _ in range(100)]
    env.step(acts[0])
    t0 = time.perf_counter()
    for k in range(1, 100):
        env.step(acts[k])
    one = 99 / (time.perf_counter() - t0)
    env.close()
    n = 256
    me = CDAVecMultiAgentEnv(dict(CFG, max_step=1000), num_envs=n)
    me.reset(seed=1)
    batch = [[_rand_dict(rng, me.agents) for _ in range(n)] for _ in range(6)]
    me.step(batch[0])
    t0 = time.perf_counter()
    for k in range(1, 6):
        me.step(batch[k])
    many = 5 * n / (time.perf_counter() - t0)
    me.close()
    with capsys.disabled():
        print(f"\n[facade] CDAEnv {one:,.0f} env-steps/s; CDAVecMultiAgentEnv(256) {many:,.0f} env-steps/s (dict protocol); reference 2,260")
    assert one > 0 and many > 0


@pytest.mark.parametrize("name", ["perm_s91", "permshuf8_s94"])
def test_dict_facade_walks_the_action_dict_in_its_iteration_order(name):
    """CDAEnv.step(action_dict): the reference draws one normal per key IN THE DICT'S ORDER and shuffles the arrival list built
    in that order (exchg/action_helper.py:164-170).  Golden traces the reference produced from dicts in a non-ascending (perm)
    or per-step shuffled (permshuf) key order, replayed through the dict facade with the dicts rebuilt in the recorded order:
    observations, rewards and env.LOB_actions (ids in dict order) match; with sorted keys they do not."""
    import golden_util as G
    from gym_continuousdoubleauction_amd import CDAEnv
    rec = G.load(name)
    A, T = rec["cat"].shape[1], 60

    def run(sort_keys):
        env = CDAEnv(rec["config"])
        env.reset(seed=int(rec["seed"]))
        for t in range(T):
            pres = rec["present"][t]
            order = sorted((a for a in range(A) if pres[a]), key=(lambda a: a) if sort_keys else (lambda a: pres[a]))
            acts = {f"agent_{a}": {"category": np.int64(rec["cat"][t, a]), "size_mean": np.array([rec["mean"][t, a]], np.float32),
                                   "size_sigma": np.array([rec["sigma"][t, a]], np.float32), "price": np.int64(rec["price"][t, a]),
                                   "price_offset": np.int64(rec["off"][t, a])} for a in order}
            obs, rew, *_ = env.step(acts)
            if not np.array_equal(obs["agent_0"].view(np.uint32), rec["obs"][t].view(np.uint32)):
                env.close()
                return t
            assert [x["ID"] for x in env.LOB_actions] == [f"agent_{a}" for a in order if rec["dec_type"][t, a] != -9]
            assert all(np.float64(rew[f"agent_{a}"]).view(np.uint64) == rec["reward"][t, a].view(np.uint64) for a in range(A))
        env.close()
        return T
    assert run(False) == T
    assert run(True) < T


def test_groups_step_batch_is_stream_ordered_like_any_other_op():
    """with groups > 1 the launches go to the group streams; by default a step is ordered after the caller's stream
    (the action tensors it just wrote) and the caller's stream after the step.  A policy-in-the-loop pattern - actions computed
    on the caller's stream right before step_batch, outputs consumed right after - equals groups = 1 bit for bit."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecMultiAgentEnv
    n, a = 2048, 4
    cfg = dict(CFG, max_step=400)
    g1, g4 = CDAVecMultiAgentEnv(cfg, num_envs=n, groups=1), CDAVecMultiAgentEnv(cfg, num_envs=n, groups=4)
    o1, _ = g1.reset_batch(seed=7)
    o4, _ = g4.reset_batch(seed=7)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    acc1 = torch.zeros(n, dtype=torch.float64, device="cuda")
    acc4 = torch.zeros_like(acc1)
    for t in range(60):
        # fresh tensors every step (the allocator recycles them), derived from the previous observation like a policy's would be
        seedish = (o1["agent_0"][:, -2:].abs().sum(dim=1, keepdim=True) * 1000).to(torch.int64)
        cat = ((torch.randint(0, 9, (n, a), generator=gen, device="cuda") + seedish) % 9).to(torch.int32)
        batch = {"category": cat, "size_mean": torch.rand((n, a), generator=gen, device="cuda") * 2 - 1,
                 "size_sigma": torch.rand((n, a), generator=gen, device="cuda"),
                 "price": torch.randint(0, 10, (n, a), generator=gen, device="cuda", dtype=torch.int32),
                 "price_offset": torch.randint(0, 3, (n, a), generator=gen, device="cuda", dtype=torch.int32)}
        o1, r1, *_ = g1.step_batch({k: v.clone() for k, v in batch.items()})
        o4, r4, *_ = g4.step_batch(batch)
        del batch, cat
        junk = torch.full((n, a), 7, dtype=torch.int32, device="cuda")          # would land in a just-freed action tensor
        acc1 += r1["agent_0"]; acc4 += r4["agent_0"]                             # consumed on the caller's stream, no explicit join
        assert torch.equal(o1["agent_0"], o4["agent_0"]), t
        del junk
    assert torch.equal(acc1, acc4)
    assert (g4.vec.flags() == 0).all()
    g1.close(); g4.close()


@pytest.mark.parametrize("auto_reset", [False, True])
def test_host_resident_step_io_equals_the_staged_copies(auto_reset, monkeypatch):
    """The facades' default step I/O - the step kernel reads the staged actions from and writes obs | reward | flags | info into pinned host memory (CDAVecEnv.bind_host_io) -
    against the staged path (CDA_FACADE_HOST_IO=0: one H2D copy of the actions, one D2H copy of `packed`): the same dicts step for step, through episode ends with the
    in-kernel auto reset (the restarted markets' first observations land in the host block too), subsets of agents and shuffled dict orders; CDAEnv likewise."""
    from gym_continuousdoubleauction_amd import CDAEnv, CDAVecMultiAgentEnv
    cfg = dict(CFG, max_step=6, auto_reset=auto_reset)
    n = 9
    monkeypatch.setenv("CDA_FACADE_HOST_IO", "1")
    a, one_a = CDAVecMultiAgentEnv(cfg, num_envs=n), CDAEnv(dict(CFG, max_step=40))
    monkeypatch.setenv("CDA_FACADE_HOST_IO", "0")
    b, one_b = CDAVecMultiAgentEnv(cfg, num_envs=n), CDAEnv(dict(CFG, max_step=40))
    assert a._host_io and one_a._host_io and not b._host_io and not one_b._host_io
    oa, _ = a.reset(seed=5); ob, _ = b.reset(seed=5)
    assert all(np.array_equal(x["agent_0"], y["agent_0"]) for x, y in zip(oa, ob))
    o1, _ = one_a.reset(seed=77); o2, _ = one_b.reset(seed=77)
    assert np.array_equal(o1["agent_0"], o2["agent_0"])
    rng = np.random.default_rng(8)
    for t in range(16 if auto_reset else 6):
        dicts = [_rand_dict(rng, a.agents) for _ in range(n)]
        if t % 4 == 1:
            del dicts[2]["agent_1"]
        if t % 4 == 2:
            dicts[3] = dict(reversed(list(dicts[3].items())))
        ra, rb = a.step(dicts), b.step(dicts)
        for i in range(n):
            assert np.array_equal(ra[0][i]["agent_0"], rb[0][i]["agent_0"]) and ra[1][i] == rb[1][i] and ra[2][i] == rb[2][i] and ra[3][i] == rb[3][i], (t, i)
            assert json.dumps(ra[4][i], sort_keys=True) == json.dumps(rb[4][i], sort_keys=True), (t, i)
        if auto_reset and t >= 6:
            assert any(tr["__all__"] for tr in ra[3]) or t % 6 != 5                     # episodes end (max_step 6) and the markets restart in place
        s1, s2 = one_a.step(dicts[0]), one_b.step(dicts[0])
        assert np.array_equal(s1[0]["agent_0"], s2[0]["agent_0"]) and s1[1] == s2[1] and s1[2] == s2[2] and s1[3] == s2[3]
        assert json.dumps(s1[4], sort_keys=True) == json.dumps(s2[4], sort_keys=True)
        assert one_a.LOB_actions == one_b.LOB_actions and one_a.pass_agents == one_b.pass_agents and one_a.done_set == one_b.done_set
    for e in (a, b, one_a, one_b):
        e.close()
