"""GPU: the single-market facade `CDAEnv` behaves like the reference env at the dict-shaped API level.
Each test restates (not copies) what the named reference test pins:
  test_env_lifecycle.py, test_observation_history.py, test_seeding.py, test_info_dict.py,
  test_new_action_space.py, test_obs_market_features.py, test_type_policy.py, test_reward_logic.py."""
import math
from decimal import Decimal

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 120, "is_render": False}


def make(cfg=None):
    from gym_continuousdoubleauction_amd import CDAEnv
    return CDAEnv(dict(cfg or CFG))


def act(cat, mean=0.1, sigma=0.0, price=0, off=1):
    return {"category": np.int64(cat), "size_mean": np.array([mean], np.float32), "size_sigma": np.array([sigma], np.float32),
            "price": np.int64(price), "price_offset": np.int64(off)}


def random_actions(env, rng):
    out = {}
    for a in env.agents:
        out[a] = {"category": np.int64(rng.integers(0, 9)), "size_mean": rng.uniform(-1, 1, 1).astype(np.float32),
                  "size_sigma": rng.uniform(0, 1, 1).astype(np.float32), "price": np.int64(rng.integers(0, 10)),
                  "price_offset": np.int64(rng.integers(0, 3))}
    return out


def test_surface_matches_the_reference_env():
    env = make({"is_render": False})        # bare env: the standalone defaults of config/env_defaults.json
    assert env.agents == [f"agent_{i}" for i in range(5)] and env.possible_agents == env.agents
    assert env._agent_ids == set(env.agents) and env.max_step == 64 and env.n_hist == 4
    assert env.observation_spaces["agent_0"].shape == (168,)
    sp = env.action_spaces["agent_3"]
    assert sp is env.action_spaces["agent_0"]                       # ONE shared Dict object
    assert sp["category"].n == 9 and sp["price"].n == 10 and sp["price_offset"].n == 3
    assert env.get_action_space("agent_1") is sp and env.get_observation_space("agent_1").shape == (168,)
    obs, infos = env.reset(seed=5)
    assert set(obs) == set(env.agents) and all(infos[a] == {} for a in env.agents)
    assert obs["agent_0"].dtype == np.float32 and obs["agent_0"].shape == (168,)
    env.close()


def test_truncation_exactly_on_max_step():                          # test_env_lifecycle.py:94-122
    for ms in (1, 2, 5, 10):
        env = make(dict(CFG, max_step=ms))
        env.reset(seed=1)
        rng = np.random.default_rng(0)
        for t in range(ms):
            _, _, term, trunc, _ = env.step(random_actions(env, rng))
            assert trunc["__all__"] == (t + 1 >= ms) and term["__all__"] is False
            assert all(term[a] is False and trunc[a] is False for a in env.agents)
        env.close()


def test_reset_pads_history_and_window_slides_and_obs_is_shared():   # test_observation_history.py
    env = make()
    obs, _ = env.reset(seed=3)
    o = obs["agent_0"]
    assert all(obs[a] is o for a in env.agents)                      # the same array object for every agent
    frames = o.reshape(4, 42)
    assert all(np.array_equal(frames[0], frames[i]) for i in range(4))
    assert frames[0][40] == np.float32(math.log(env.last_price)) and frames[0][41] == 0.0 and not frames[0][:40].any()
    prev = o.copy()
    rng = np.random.default_rng(1)
    for _ in range(6):
        obs, *_ = env.step(random_actions(env, rng))
        cur = obs["agent_1"]
        assert np.array_equal(cur[: 3 * 42], prev[42:])              # sliding window: oldest frame dropped
        assert not np.isnan(cur).any()
        prev = cur.copy()
    env.close()


def _trajectory(seed, steps=60):
    env = make()
    env.reset(seed=seed)
    rng = np.random.default_rng(7)                                   # the actions are held fixed across runs
    rec = []
    for _ in range(steps):
        _, rewards, _, _, infos = env.step(random_actions(env, rng))
        rec.append((env.last_price, tuple(rewards.values()), tuple(infos[a]["NAV"] for a in env.agents),
                    tuple(infos[a]["net_position"] for a in env.agents)))
    env.close()
    return rec


def test_reset_seed_is_honoured():                                   # test_seeding.py:43-105
    assert _trajectory(123) == _trajectory(123)
    assert _trajectory(123) != _trajectory(456)
    env = make()
    env.reset(seed=9)
    p1 = env.last_price
    env.reset(seed=9)
    assert env.last_price == p1
    anchors = set()
    for _ in range(12):                                              # seed=None continues the stream
        env.reset()
        anchors.add(env.last_price)
        assert 10 <= env.last_price <= 100
    assert len(anchors) > 1
    env.close()


def test_info_dict_fields_types_and_reward_decomposition():         # test_info_dict.py
    env = make()
    env.reset(seed=2)
    rng = np.random.default_rng(3)
    saw_trade = False
    for _ in range(80):
        acts = random_actions(env, rng)
        _, rewards, _, _, infos = env.step(acts)
        for a in env.agents:
            i = infos[a]
            assert set(i) >= {"reward", "NAV", "num_trades", "net_position", "VWAP", "cash", "cash_on_hold", "position_val",
                              "drawdown", "max_nav", "num_trades_step", "num_passive_fills_step", "order_step_placed",
                              "num_rejected_step", "is_pass_action", "reward_terms", "last_price", "best_bid", "best_ask",
                              "spread", "model_action"}
            assert isinstance(i["NAV"], str) and str(Decimal(i["NAV"])) == i["NAV"]
            assert type(i["net_position"]) is int and type(i["num_trades"]) is int and type(i["is_pass_action"]) is bool
            assert i["reward"] == rewards[a]
            total = 0.0
            for v in i["reward_terms"].values():                     # left to right, bit exact (info_helper / reward_helper:92-94)
                total += v
            assert total == i["reward"]
            assert list(i["reward_terms"]) == ["nav_term", "order_penalty", "trade_penalty", "drawdown_penalty", "passive_bonus"]
            assert i["is_pass_action"] == (int(acts[a]["category"]) == 0)
            assert (i["spread"] is None) == (i["best_bid"] is None or i["best_ask"] is None)
            if i["spread"] is not None:
                assert i["spread"] == i["best_ask"] - i["best_bid"] > 0
            saw_trade |= i["num_trades_step"] > 0
    assert saw_trade
    assert sum(Decimal(infos[a]["NAV"]) for a in env.agents) == pytest.approx(Decimal(4000000), abs=Decimal("1e-15"))
    env.close()


def test_price_ladder_offsets_and_pass_action():                    # test_new_action_space.py:32-140
    env = make()
    env.reset(seed=4)
    anchor = int(env.last_price)
    # empty book: ghost ladder anchor -/+ (level+1)*tick, offsets -1/0/+1 tick with the bid/ask sign
    env.step({"agent_0": act(2, price=2, off=1)})                   # bid, level 2, join
    bids, asks = env.book()
    assert bids[0]["price"] == anchor - 3 and not asks
    env.step({"agent_1": act(6, price=0, off=2)})                   # ask, level 0, aggressive = one tick lower
    bids, asks = env.book()
    assert asks[0]["price"] == max(anchor + 1 - 1, bids[0]["price"] + 1) or asks == [] or asks[0]["price"] == anchor
    raw = env.agg_LOB_raw
    assert raw.shape == (40,) and raw.dtype == np.float32 and raw[0] == anchor - 3
    before = env.book()
    _, _, _, _, infos = env.step({"agent_2": act(0)})               # category 0: no LOB action
    assert env.book() == before and infos["agent_2"]["is_pass_action"] is True and set(infos) == set(env.agents)
    # a subset of agents may act; absent agents still get obs/reward/info
    obs, rewards, _, _, infos = env.step({"agent_3": act(2, price=0, off=0)})
    assert set(obs) == set(rewards) == set(infos) == set(env.agents) and "model_action" not in infos["agent_0"]
    env.close()


def test_market_features_log_mid_and_spread():                      # test_obs_market_features.py:76-178
    env = make()
    env.reset(seed=6)
    anchor = int(env.last_price)
    obs, *_ = env.step({"agent_0": act(2, price=0, off=1)})          # one-sided (bid only): M = best bid, spread term 0
    f = obs["agent_0"][-42:]
    assert f[40] == np.float32(math.log(anchor - 1)) and f[41] == 0.0 and f[0] == 0.0 and f[10] > 0
    obs, *_ = env.step({"agent_1": act(6, price=2, off=1)})          # ask at anchor + 3 -> two-sided
    f = obs["agent_0"][-42:]
    bid, ask = anchor - 1, anchor + 3
    M = (bid + ask) / 2.0
    assert f[40] == np.float32(math.log(M)) and f[41] == np.float32(math.log1p(ask - bid))
    assert f[0] == np.float32((M - bid) / M) and f[20] == np.float32(-((ask - M) / M)) and f[30] < 0
    env.close()


def test_bad_inputs_raise_like_the_reference():
    env = make()
    env.reset(seed=1)
    with pytest.raises(KeyError):
        env.step({"agent_0": act(11)})                               # _CATEGORY_MAP[11] -> KeyError
    with pytest.raises(ValueError):
        env.step({"agent_0": act(2, sigma=-0.5)})                    # numpy: scale < 0
    env.close()


def test_random_agent_driver_runs_to_the_horizon():                  # CDA_rand.run_random / BASELINE config #1
    from gym_continuousdoubleauction_amd import run_random
    assert run_random(num_agents=4, max_step=200, seed=123) == 200


def test_handback_records_rebuild_the_full_outputs_on_the_receiving_side():
    """The multi-GPU hand-back with one rank on the real kernels: k_step / k_reset write one compact record per market (newest frame |
    reward | flags), every group chain moves its own range and cda_handback_unpack rebuilds the learner-side arrays - which
    must equal the outputs of a plain env step for step, through a mid-run reset and with auto_reset (records flagged
    `restarted`).  The unpack kernel is also compared with the numpy restatement the gloo tests use."""
    import numpy as np
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd.parallel import ShardedVecEnv
    from test_distributed_gloo import unpack_restated
    n, a = 200, 4
    for auto, force in ((False, False), (True, False), (False, True)):
        cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 12, "is_render": False, "auto_reset": auto}
        ref = CDAVecEnv(cfg, n_markets=n, with_info=False)
        # world 1.  force: every chain gets a one-rank RCCL communicator and the native step really calls ncclAllGather on the chain's
        # stream (RCCL loaded and resolved from the process, unique id, ncclCommInitRank, the gathered buffer as the unpack's source)
        sh = ShardedVecEnv(cfg, n, device="cuda:0", groups=4, handback=True, force_collective=force)
        assert len(sh.group_ranges) == 4 and sh.env.handback.shape == (n, 208) and sh.transport == "rccl" and (sh._comms is not None) == force
        o0 = ref.reset(seed=np.arange(300, 300 + n, dtype=np.uint64)).clone()
        assert torch.equal(sh.reset(seed_base=300), o0) and torch.equal(sh.full[0], o0)
        rng = np.random.default_rng(3)
        shadow = [x.cpu().clone() for x in sh.full]                                    # the restated receiving side, fed the same records
        for t in range(40):
            acts = (torch.from_numpy(rng.integers(0, 9, (n, a)).astype(np.int32)), torch.from_numpy(rng.uniform(-1, 1, (n, a)).astype(np.float32)),
                    torch.from_numpy(rng.uniform(0, 1, (n, a)).astype(np.float32)), torch.from_numpy(rng.integers(0, 10, (n, a)).astype(np.int32)),
                    torch.from_numpy(rng.integers(0, 3, (n, a)).astype(np.int32)))
            ro, rr, rt, ru, _ = ref.step(*acts)
            so, sr, st, su, _ = sh.step(*acts)
            assert torch.equal(so.view(torch.int32), ro.view(torch.int32)) and torch.equal(sr.view(torch.int64), rr.view(torch.int64))
            fo, fr, ft, fu = sh.full
            assert torch.equal(fo.view(torch.int32), ro.view(torch.int32)), (auto, t)
            assert torch.equal(fr.view(torch.int64), rr.view(torch.int64)) and torch.equal(ft != 0, rt) and torch.equal(fu != 0, ru)
            unpack_restated(sh.env.handback.cpu(), 1, n, n, 0, a, 4, *shadow)
            assert all(torch.equal(x.cpu().view(torch.uint8), y.view(torch.uint8)) for x, y in zip(sh.full, shadow)), (auto, t)
            if not auto and t in (11, 23):                                             # episodes end at max_step = 12: reset everybody
                o = ref.reset().clone()
                assert torch.equal(sh.reset(seed_base=None), o) and torch.equal(sh.full[0], o)
                unpack_restated(sh.env.handback.cpu(), 1, n, n, 0, a, 4, *shadow)
        ref.close()
        sh.close()


def test_ppo_loop_on_the_hip_env_with_graph_captured_policy_step():
    """BASELINE config #5 in miniature: rollouts through a captured HIP graph of the policy step, device-side auto reset."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd import ppo
    env = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 12, "is_render": False, "auto_reset": True},
                    n_markets=256, with_info=False)
    logs = []
    model, hist = ppo.train(env, iters=2, horizon=16, log=logs.append)
    assert not any("capture failed" in x for x in logs), logs
    assert len(hist) == 2 and all(math.isfinite(h["pg_loss"]) and math.isfinite(h["v_loss"]) and h["agent_steps"] == 256 * 4 * 16 for h in hist)
    assert int(env.flags().abs().sum()) == 0
    eager_model, eager_hist = ppo.train(env, iters=1, horizon=4, log=logs.append, use_graph=False)
    assert math.isfinite(eager_hist[0]["v_loss"])
    env.close()
    del model, eager_model
    torch.cuda.synchronize()


def test_whole_rollout_step_graph_with_market_groups_and_graphed_update():
    """The rollout step captured as ONE HIP graph (policy, env step on TWO group streams forked from / joined into the capturing
    stream, device-side auto reset, buffer writes at a device-side step index) and the update's minibatch steps replayed from graphs
    from the second iteration on: the loop learns on what the env produced - the buffers hold T distinct steps (the step index moved),
    the recorded actions are the ones the env consumed (its own trade counters move), nothing is flagged - and the per-sample variant
    runs through the same machinery."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 12, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=256, with_info=False, groups=2)
    logs = []
    model, hist = ppo.train(env, iters=3, horizon=16, log=logs.append)
    assert not any("capture failed" in x for x in logs), logs
    assert len(hist) == 3 and all(math.isfinite(h["pg_loss"]) and math.isfinite(h["v_loss"]) and math.isfinite(h["entropy"]) for h in hist)
    assert hist[1]["v_loss"] != hist[2]["v_loss"]                       # (the graphed updates see fresh rollouts)
    assert int(env.flags().abs().sum()) == 0 and int((env.check_invariants() != 0).sum()) == 0
    model2, hist2 = ppo.train(env, iters=2, horizon=8, log=logs.append, shared_obs=False)
    assert all(math.isfinite(h["v_loss"]) for h in hist2) and not any("capture failed" in x for x in logs), logs
    # the graph's buffers against an eager replay of the same policy draws: rebuild the step by hand for one fresh env
    env2 = CDAVecEnv(cfg, n_markets=64, with_info=False)
    env2.reset(seed=5)
    m = ppo.ActorCritic(env2.obs_dim).to(env2.device)
    g, buf, t_dev = ppo._capture_rollout_step(m, env2, 64, 4, 6, seed=9, shared=True)
    env2.reset(seed=5)
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
    assert int(t_dev.item()) == 6
    ref = CDAVecEnv(cfg, n_markets=64, with_info=False)
    ref.reset(seed=5)
    st = ppo.new_sampler_state(9, ref.device)
    st[1].fill_(3)                                                       # the capture's three warm-up draws moved the counter
    for t in range(6):
        assert torch.equal(buf["obs"][t], ref.obs)
        with torch.no_grad():
            acts, logp, val, env_acts = m.act_fused(ref.obs, 64, 4, st, shared=True)
        assert torch.equal(buf["a_cat"][t], acts[0]) and torch.equal(buf["a_cont"][t], acts[3]) and torch.equal(buf["logp"][t], logp)
        _, r, term, trunc, _ = ref.step(*env_acts)
        assert torch.equal(buf["rew"][t], r) and torch.equal(buf["term"][t], term) and torch.equal(buf["trunc"][t], trunc)
    assert torch.equal(env2.obs, ref.obs)
    env.close(); env2.close(); ref.close()


def test_graph_replayed_update_equals_the_eager_update():
    """ppo_update(graphs=...): the first call runs eagerly and captures one HIP graph per minibatch step; later calls replay them.  Two
    copies of one model, one always eager, one through the graphs, fed the same data under the same torch seed (= the same epoch
    permutations): the parameters must stay together - through the capturing call exactly as eager as the other, and through two
    replayed updates to float32 / bfloat16 rounding of differently ordered atomic sums (log_std's gradient comes from atomics)."""
    import copy
    import torch
    from gym_continuousdoubleauction_amd import ppo
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    R, A = 8192, 4
    m1 = ppo.ActorCritic(168).to(dev)
    m2 = copy.deepcopy(m1)
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-3, fused=True, capturable=True)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, fused=True, capturable=True)
    graphs = {}
    for call in range(3):
        obs = torch.randn(R, 168, device=dev)
        with torch.no_grad():
            acts, logp_old, _ = m1.act(obs.repeat_interleave(A, dim=0))
        adv, ret = torch.randn(R * A, device=dev), torch.randn(R * A, device=dev)
        stats = []
        for m, o, g in ((m1, o1, None), (m2, o2, graphs)):
            torch.manual_seed(100 + call)
            stats.append(ppo.ppo_update(m, o, obs, acts, logp_old, adv, ret, epochs=2, minibatch=4096 * A // 2, amp=True, agents_per_row=A, graphs=g))
        assert ("update" in graphs) and len(graphs["update"].graphs) == 4          # 8192 rows in minibatches of 2048 rows
        worst = 0.0
        with torch.no_grad():
            for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
                # log_std's gradient is a sum of atomics: when it is near zero its rounding can flip Adam's normalised step (lr = 1e-3 whatever the
                # gradient's size), so that one parameter is held to a few steps; everything else to float32 / bfloat16 rounding
                tol = 3e-3 if n1 == "log_std" else 2e-4
                if n1 != "log_std":
                    worst = max(worst, float((p1 - p2).abs().max()))
                assert torch.allclose(p1, p2, rtol=0, atol=tol), (call, n1, float((p1 - p2).abs().max()))
        assert abs(stats[0]["v_loss"] - stats[1]["v_loss"]) < 1e-3 * max(1.0, abs(stats[0]["v_loss"])), (call, stats)
    assert worst < 2e-4


def test_fused_gae_kernel_equals_the_recursion():
    import torch
    from gym_continuousdoubleauction_amd import ppo
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    T, B = 37, 5000
    rew, val, last = torch.randn(T, B, device=dev), torch.randn(T, B, device=dev), torch.randn(B, device=dev)
    done = (torch.rand(T, B, device=dev) < 0.1).float()
    adv, ret = ppo.gae(rew, val, last, done)
    adv_ref, ret_ref = ppo.gae(rew.double(), val.double(), last.double(), done.double(), fused=False)
    assert float((adv.double() - adv_ref).abs().max()) < 1e-4 and float((ret.double() - ret_ref).abs().max()) < 1e-4


def test_league_rollout_routes_modules_per_market():
    """League slot mapping on the batched env: a trainable policy in slot 0, fixed random opponents and a champion elsewhere."""
    import numpy as np
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd.league import LeagueSlotMapper, RandomModule, league_actions
    n, a = 128, 4
    env = CDAVecEnv({"num_of_agents": a, "init_cash": 1000000, "max_step": 64, "is_render": False}, n_markets=n, with_info=True)
    mapper = LeagueSlotMapper(a, 1, 3, original_opponent_weight=1.0, champion_weight=3.0)
    champ = mapper.add_champion()
    assignment = mapper.assign([f"iter0-market{i}" for i in range(n)])
    always_pass = lambda o: (torch.zeros(o.shape[0], dtype=torch.int32, device=o.device), torch.zeros(o.shape[0], device=o.device),   # noqa: E731
                             torch.zeros(o.shape[0], device=o.device), torch.zeros(o.shape[0], dtype=torch.int32, device=o.device),
                             torch.ones(o.shape[0], dtype=torch.int32, device=o.device))
    modules = {name: RandomModule("cuda:0", seed=i) for i, name in enumerate(mapper.available_modules)}
    modules[champ] = always_pass                               # the champion never trades: recognisable in the info tensors
    obs = env.reset(seed=7)
    for t in range(20):
        obs, rew, term, trunc, info = env.step(*league_actions(mapper, assignment, modules, obs))
    names = mapper.names(assignment)
    is_champ = torch.as_tensor(names == champ, device="cuda:0")
    assert bool(is_champ.any()) and bool((~is_champ).any())
    assert bool(info["is_pass_action"].bool()[is_champ].all())              # last step: every champion slot passed
    assert int(info["num_trades"][is_champ].sum()) == 0 and int(info["num_trades"][~is_champ].sum()) > 0
    assert int(env.flags().abs().sum()) == 0
    env.close()


def test_lob_actions_hold_the_decoded_orders():              # test_new_action_space.py:36-147, test_obs_normalization.py:314-338
    from gym_continuousdoubleauction_amd import CDAEnv
    env = CDAEnv({"num_of_agents": 3, "init_cash": 1000000, "max_step": 32, "is_render": False})
    env.reset(seed=5)
    assert env.LOB_actions is None
    lp = env.last_price
    one = lambda c, price=0, off=1: {"category": c, "size_mean": np.array([0.0], np.float32), "size_sigma": np.array([0.0], np.float32),   # noqa: E731
                                     "price": price, "price_offset": off}
    env.step({"agent_1": one(2, price=3, off=1)})           # bid limit on an empty book: last_price - (level + 1) ticks
    assert len(env.LOB_actions) == 1
    act = env.LOB_actions[0]
    assert act == {"ID": "agent_1", "side": "bid", "type": "limit", "size": 1, "price": lp - 4.0}
    env.step({"agent_0": one(5), "agent_1": one(0), "agent_2": one(8, price=0, off=0)})
    got = {a["ID"]: a for a in env.LOB_actions}
    assert set(got) == {"agent_0", "agent_2"}                # the passing agent is left out
    assert got["agent_0"]["type"] == "market" and got["agent_0"]["side"] == "ask" and got["agent_0"]["price"] == -1.0
    assert got["agent_2"]["type"] == "cancel" and got["agent_2"]["side"] == "ask" and got["agent_2"]["price"] > 0
    env.reset(seed=5)
    assert env.LOB_actions is None
    # keys that only RESOLVE to an agent ("agent_02" -> agent 2: _encode's fallback) address that agent in every output too
    _, _, _, _, infos = env.step({"agent_02": one(2, price=1), "agent_0": one(1)})
    assert [a["ID"] for a in env.LOB_actions] == ["agent_2", "agent_0"]          # dict order, canonical ids
    assert infos["agent_2"]["model_action"] is not None and infos["agent_0"]["model_action"] is not None and "model_action" not in infos["agent_1"]
    env.close()


def test_reward_terms_follow_the_reference_formulas_exactly():      # test_reward_logic.py:15-110
    """Every term recomputed from the exact NAV strings: nav_term = float(dNAV) * (1.5 if negative), the high-water mark
    and drawdown, the three counter terms - bit for bit, over a random episode with trades, losses and passive fills."""
    env = make()
    env.reset(seed=12)
    rng = np.random.default_rng(5)
    prev = {a: Decimal(1000000) for a in env.agents}
    high = dict(prev)
    saw_loss = saw_gain = saw_passive = saw_drawdown = False
    for _ in range(120):
        _, rewards, _, _, infos = env.step(random_actions(env, rng))
        for a in env.agents:
            i, t = infos[a], infos[a]["reward_terms"]
            nav = Decimal(i["NAV"])
            change = float(nav - prev[a])
            assert t["nav_term"] == change * (1.5 if change < 0 else 1.0)
            high[a] = max(high[a], nav)                              # max_nav is a high-water mark of the exact NAV
            assert i["max_nav"] == float(high[a])
            dd = float(max(Decimal(0), high[a] - nav))
            assert i["drawdown"] == dd and t["drawdown_penalty"] == -(0.2 * dd)
            assert t["order_penalty"] == -(0.1 * i["order_step_placed"]) and t["trade_penalty"] == -(0.05 * i["num_trades_step"])
            assert t["passive_bonus"] == 0.1 * i["num_passive_fills_step"] and i["num_passive_fills_step"] <= i["num_trades_step"]
            saw_loss |= change < 0
            saw_gain |= change > 0
            saw_passive |= i["num_passive_fills_step"] > 0
            saw_drawdown |= dd > 0
            prev[a] = nav
    assert saw_loss and saw_gain and saw_passive and saw_drawdown
    env.close()


def test_observation_normalisation_rules():                          # test_obs_normalization.py:60-340
    env = make()
    obs, _ = env.reset(seed=9)
    anchor = int(env.last_price)
    f = obs["agent_0"][-42:]
    assert env.agg_LOB_raw.shape == (40,) and not env.agg_LOB_raw.any()                    # empty book after reset
    assert not f[:40].any() and f[40] == np.float32(math.log(anchor)) and f[41] == 0.0     # last_price is the anchor
    sized = lambda c, mean, price, off=1: {"category": c, "size_mean": np.array([mean], np.float32), "size_sigma": np.array([0.0], np.float32),   # noqa: E731
                                           "price": price, "price_offset": off}
    # limit sizes: rint(|lim_mul * mean|) + min_size with lim_mul = (100 * 10 - 1) / 2 = 499.5
    obs, *_ = env.step({"agent_0": sized(2, 0.1, 0), "agent_1": sized(2, 0.2, 2), "agent_2": sized(6, 0.05, 1)})
    f = obs["agent_0"][-42:]
    raw = env.agg_LOB_raw
    bid0, bid1, ask0 = anchor - 1, anchor - 3, anchor + 2
    size_of = lambda mean: int(np.rint(abs(np.float32(499.5) * np.float32(mean)))) + 1    # noqa: E731
    s0, s1, sa = size_of(0.1), size_of(0.2), size_of(0.05)
    assert list(raw[:2]) == [bid0, bid1] and list(raw[10:12]) == [s0, s1] and raw[20] == -ask0 and raw[30] == -sa
    M = (bid0 + ask0) / 2.0
    assert f[0] == np.float32((M - bid0) / M) and f[1] == np.float32((M - bid1) / M) and (f[:10] >= 0).all()      # bids >= 0
    assert f[20] == np.float32(-((ask0 - M) / M)) and (f[20:30] <= 0).all()                                        # asks <= 0
    assert f[10] == np.float32(math.sqrt(s0)) and f[11] == np.float32(math.sqrt(s1)) and f[30] == np.float32(-math.sqrt(sa))
    assert f[0] == -f[20]                                                                  # level-1 distances are symmetric about the mid
    assert all(a["price"] > 0 for a in env.LOB_actions)                                    # resolved action prices are raw and positive
    # a zero last_price on an empty book falls back to M = 100
    env2 = make()
    env2.reset(seed=9)
    env2.last_price = 0
    obs2, *_ = env2.step({"agent_0": sized(0, 0.0, 0)})
    assert obs2["agent_0"][-42:][40] == np.float32(math.log(100.0))
    env.close(); env2.close()


def test_league_self_play_loop_on_the_hip_env():
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd.league_train import train_league
    import tempfile
    import pyarrow.parquet as pq
    from gym_continuousdoubleauction_amd.episode_record import BatchedEpisodeRecorder
    env = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 16, "is_render": False}, n_markets=256, with_info=True)
    out = tempfile.mkdtemp()
    rec = BatchedEpisodeRecorder(out, num_agents=4, markets=[3, 200], run_id="league-test")
    model, mapper, hist = train_league(env, iters=3, promote_margin=-1e9, recorder=rec, log=lambda s: None)
    rec.close()
    t = pq.read_table(rec.files).to_pandas()
    assert len(t) == 3 * 2 * 16 * 4 and set(t["iteration"]) == {0, 1, 2} and t["episode_complete"].all()
    assert (t[t.agent_id == "agent_0"]["module_id"] == "policy_0").all()                   # the trainable slot
    late = t[(t.iteration == 2) & (t.agent_id != "agent_0")]["module_id"]
    assert set(late) <= {"policy_1", "policy_2", "policy_3", "champion_1", "champion_2"}   # champion_3 joins after iteration 2
    assert [h["promoted"] for h in hist] == ["champion_1", "champion_2", "champion_3"]
    assert all(math.isfinite(h["v_loss"]) and math.isfinite(h["episode_return"]) for h in hist)
    assert int(env.flags().abs().sum()) == 0
    env.close()


@pytest.mark.parametrize("per_row", [1, 4, 16])
def test_fused_ppo_loss_kernel_equals_the_pytorch_statement(per_row):
    """cda_ppo_loss (csrc/cda_ppo.hip): loss, its three means and the gradients with respect to logits, value and log_std of one
    minibatch in one launch - against the plain PyTorch fp32 formulation of the same op (ActorCritic.evaluate + the loss formulas
    of ppo_update) on the same inputs.  float32 tolerance: 2e-5 relative on the scalars, 1e-6 absolute on the gradients (they carry
    the 1 / B of the means; B = 20 000 samples here).
    per_row > 1: a row of network outputs serves that many consecutive samples (the agents of a market share the observation); the
    PyTorch statement is the plain per-sample one on the REPLICATED rows, whose gradient with respect to a row autograd sums."""
    import torch
    from gym_continuousdoubleauction_amd import ppo
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    B = 20000
    R = B // per_row
    m = ppo.ActorCritic(168).to(dev)
    with torch.no_grad():
        m.log_std.copy_(torch.tensor([-0.3, 0.2]))
    obs = torch.randn(R, 168, device=dev)
    with torch.no_grad():
        acts, logp_old, _ = m.act(obs.repeat_interleave(per_row, dim=0))
        logp_old = logp_old + 0.3 * torch.randn_like(logp_old)          # ratios on both sides of the clip range
    adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
    clip, vf, ec = 0.2, 0.5, 0.01
    # the reference: autograd through the PyTorch formulation, gradients taken at the network OUTPUTS (one row per market-step)
    o, v = m.trunk(obs)
    o, v = o.detach().float().requires_grad_(True), v.detach().float().requires_grad_(True)
    ls = m.log_std.detach().clone().requires_grad_(True)
    os_, vs_ = o.repeat_interleave(per_row, dim=0), v.repeat_interleave(per_row, dim=0)
    logp = ent = 0.0
    for lo, hi, a in ((0, 9, acts[0]), (9, 19, acts[1]), (19, 22, acts[2])):
        l = torch.log_softmax(os_[:, lo:hi], dim=-1)
        logp = logp + l.gather(1, a.view(-1, 1)).squeeze(1)
        ent = ent - (l.exp() * l).sum(-1)
    z = (acts[3] - os_[:, -2:]) * torch.exp(-ls)
    logp = logp + (-0.5 * z * z - ls - 0.9189385332046727).sum(-1)
    ent = ent + (1.4189385332046727 + ls).sum()
    ratio = (logp - logp_old).exp()
    pg = -torch.min(ratio * adv, ratio.clamp(1 - clip, 1 + clip) * adv).mean()
    vl = (vs_ - ret).pow(2).mean()
    loss = pg + vf * vl - ec * ent.mean()
    loss.backward()
    o2, v2 = o.detach().clone().requires_grad_(True), v.detach().clone().requires_grad_(True)
    ls2 = ls.detach().clone().requires_grad_(True)
    loss2, out = ppo._FusedPPOLoss.apply(o2, v2, ls2, acts[0], acts[1], acts[2], acts[3].float(), logp_old.float(), adv, ret, clip, vf, ec, per_row)
    loss2.backward()
    rel = lambda a, b: float((a.detach() - b.detach()).abs() / b.detach().abs().clamp_min(1e-6))   # noqa: E731
    assert rel(loss2, loss) < 2e-5 and rel(out[0], pg) < 2e-5 and rel(out[1], vl) < 2e-5 and rel(out[2], ent.mean()) < 2e-5
    assert float((o2.grad - o.grad).abs().max()) < 1e-6 and float((v2.grad - v.grad).abs().max()) < 1e-6
    assert float((ls2.grad - ls.grad).abs().max()) < 2e-5 * max(1.0, float(ls.grad.abs().max()))
    assert float((o.grad != 0).float().mean()) > 0.5                      # (the comparison is not between zeros)
    # row_index: the rows are handed over SHUFFLED and the kernel finds each row's samples through the permutation - same loss, the
    # same gradients in the shuffled order
    perm = torch.randperm(R, device=dev)
    o3, v3 = o.detach()[perm].clone().requires_grad_(True), v.detach()[perm].clone().requires_grad_(True)
    loss3, out3 = ppo._FusedPPOLoss.apply(o3, v3, ls.detach(), acts[0], acts[1], acts[2], acts[3].float(), logp_old.float(), adv, ret, clip, vf, ec, per_row, perm)
    loss3.backward()
    assert rel(loss3, loss2) < 1e-6 and torch.equal(o3.grad, o2.grad[perm]) and torch.equal(v3.grad, v2.grad[perm])
    # packed: the network's padded output matrix [R, 32] (logits | value | zeros) in, its gradient out
    pad = torch.cat([o.detach(), v.detach().unsqueeze(1), torch.zeros(R, 7, device=dev)], dim=1).requires_grad_(True)
    loss5, out5 = ppo._FusedPPOLossPacked.apply(pad, ls.detach(), acts[0], acts[1], acts[2], acts[3].float(), logp_old.float(), adv, ret, clip, vf, ec, per_row)
    loss5.backward()
    assert torch.equal(loss5.detach(), loss2.detach()) and torch.equal(pad.grad[:, :24], o2.grad) and torch.equal(pad.grad[:, 24], v2.grad)
    assert float(pad.grad[:, 25:].abs().max()) == 0.0
    half = perm[: R // 2].contiguous()                                    # a minibatch: half of the rows, means over ITS samples
    loss4, out4 = ppo._FusedPPOLoss.apply(o.detach()[half], v.detach()[half], ls.detach(), acts[0], acts[1], acts[2], acts[3].float(), logp_old.float(), adv, ret,
                                          clip, vf, ec, per_row, half)
    sel = (half.view(-1, 1) * per_row + torch.arange(per_row, device=dev)).view(-1)
    r4 = ratio.detach()[sel]
    pg4 = -torch.min(r4 * adv[sel], r4.clamp(1 - clip, 1 + clip) * adv[sel]).mean()
    assert rel(out4[0], pg4) < 2e-5 and rel(out4[1], (vs_.detach()[sel] - ret[sel]).pow(2).mean()) < 2e-5


def test_fused_policy_sampling_kernel_follows_the_network_outputs():
    """cda_policy_sample: (i) the log-probability it returns for the action it drew equals ActorCritic.evaluate's for that action
    (the plain PyTorch fp32 formulation; 1e-4 absolute); (ii) the env tensors are the squashed / cast samples; (iii) the draws
    follow the distribution: category frequencies over 400 k rows within 4 sigma of the softmax probabilities, Gaussian samples
    with the right mean / std; (iv) consecutive calls (and graph replays) draw different numbers, equal seeds equal numbers."""
    import torch
    from gym_continuousdoubleauction_amd import ppo
    torch.manual_seed(2)
    dev = torch.device("cuda:0")
    n, a = 100000, 4
    m = ppo.ActorCritic(168).to(dev)
    with torch.no_grad():
        m.out.bias[:9] = torch.linspace(-1.5, 1.5, 9, device=dev)             # a visibly non-uniform category head
    obs = (0.05 * torch.randn(1, 168, device=dev)).expand(n * a, 168).contiguous()  # the same observation in every row
    st = ppo.new_sampler_state(11, dev)
    with torch.no_grad():
        acts, logp, val, env_acts = m.act_fused(obs, n, a, st)
        lp_ref, _, v_ref = m.evaluate(obs, acts)
    assert float((logp - lp_ref).abs().max()) < 1e-4 and float((val - v_ref).abs().max()) < 1e-5
    assert torch.equal(env_acts[0].view(-1), acts[0].to(torch.int32)) and torch.equal(env_acts[3].view(-1), acts[1].to(torch.int32))
    assert torch.equal(env_acts[4].view(-1), acts[2].to(torch.int32))
    assert float((env_acts[1].view(-1) - torch.tanh(acts[3][:, 0])).abs().max()) < 1e-6
    assert float((env_acts[2].view(-1) - torch.sigmoid(acts[3][:, 1])).abs().max()) < 1e-6
    assert int(acts[0].min()) >= 0 and int(acts[0].max()) <= 8 and int(acts[1].max()) <= 9 and int(acts[2].max()) <= 2
    with torch.no_grad():
        o, _ = m.trunk(obs[:1])
    for lo, hi, x in ((0, 9, acts[0]), (9, 19, acts[1]), (19, 22, acts[2])):
        p = torch.softmax(o[0, lo:hi].float(), -1)
        f = torch.bincount(x, minlength=hi - lo).float() / (n * a)
        assert float(((f - p).abs() / (p * (1 - p) / (n * a)).sqrt()).max()) < 4.5, (lo, f, p)
    mu, sd = o[0, -2:].float(), m.log_std.detach().exp()
    assert float(((acts[3].mean(0) - mu).abs() / (sd / (n * a) ** 0.5)).max()) < 4.5 and float((acts[3].std(0) / sd - 1).abs().max()) < 0.01
    with torch.no_grad():
        acts2, *_ = m.act_fused(obs, n, a, st)                                  # the device-side counter moved on
        acts3, *_ = m.act_fused(obs, n, a, ppo.new_sampler_state(11, dev))     # same seed, fresh counter: the first draw again
    assert not torch.equal(acts2[0], acts[0]) and torch.equal(acts3[0], acts[0]) and torch.equal(acts3[3], acts[3])
    assert int(st[1].item()) == 2
    # shared rows: the network runs on ONE row per market and the market's `a` agents draw from it - same numbers as the replicated
    # rows above (the draw is keyed by the sample, the logits are equal), value per market
    with torch.no_grad():
        acts4, logp4, val4, env4 = m.act_fused(obs[:n].contiguous(), n, a, ppo.new_sampler_state(11, dev), shared=True)
    # (the n-row and the n*a-row products may round their logits differently in the last bit: a draw that sits on a CDF boundary can flip)
    same = (acts4[0] == acts[0]) & (acts4[1] == acts[1]) & (acts4[2] == acts[2])
    assert float(same.float().mean()) > 0.9995 and float((acts4[3] - acts[3]).abs().max()) < 1e-5 and val4.shape == (n,)
    assert float((logp4 - logp)[same].abs().max()) < 1e-4 and float((val4 - val[::a]).abs().max()) < 1e-5
    assert torch.equal(env4[0].view(-1), acts4[0].to(torch.int32)) and float((env4[1].view(-1) - torch.tanh(acts4[3][:, 0])).abs().max()) < 1e-6
