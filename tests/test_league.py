"""CPU: league slot mapping vs the reference's own mapping function (golden cut by tests/golden/make_league_golden.py)
and vs numpy's legacy RandomState, the third-party generator the reference delegates the draw to."""
import json
import os
import zlib

import numpy as np

from gym_continuousdoubleauction_amd.league import LeagueSlotMapper, mt19937_first_double

HERE = os.path.dirname(os.path.abspath(__file__))


def test_first_double_matches_numpy_randomstate():
    rng = np.random.default_rng(1)
    seeds = np.concatenate([rng.integers(0, 2 ** 32, 500, dtype=np.uint64), np.array([0, 1, 2 ** 32 - 1, 2 ** 31], dtype=np.uint64)])
    got = mt19937_first_double(seeds)
    want = np.array([np.random.RandomState(int(s)).random_sample() for s in seeds])
    assert np.array_equal(got, want)


def test_assignment_equals_the_reference_mapping_function():
    with open(os.path.join(HERE, "golden", "league_mapping.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 5
    for c in cases:
        m = LeagueSlotMapper(c["num_agents"], c["num_trainable"], c["num_fixed"], c["original_opponent_weight"], c["champion_weight"])
        for ch in c["champions"]:
            m.add_champion(ch)
        assert m.available_modules == c["available_modules"]
        idx = m.assign(c["episode_ids"])
        assert idx.shape == (len(c["episode_ids"]), c["num_agents"])
        assert m.names(idx).tolist() == c["assignment"]


def test_choice_semantics_and_grouping():
    m = LeagueSlotMapper(6, 2, 4, original_opponent_weight=1.0, champion_weight=4.0)
    assert m.add_champion() == "champion_1" and m.add_champion() == "champion_2"
    ids = [f"e{i}" for i in range(300)]
    idx = m.assign(ids)
    pool, p = m.pool(), m.pool_probabilities()
    for i in (0, 7, 299):                                   # the same draw through numpy's own choice()
        for a in range(2, 6):
            seed = (zlib.crc32(ids[i].encode()) + a) % 2 ** 32
            assert m.available_modules[idx[i, a]] == str(np.random.RandomState(seed).choice(pool, p=p))
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == 1).all() and (idx[:, 2:] >= 2).all()
    groups = m.group_by_module(idx)
    assert sum(len(mk) for mk, _ in groups.values()) == idx.size
    for name, (mk, sl) in groups.items():
        assert (m.names(idx)[mk, sl] == name).all()
    champs = np.isin(idx[:, 2:], [6, 7]).mean()              # 2 champions x 4 vs 4 originals x 1  ->  2/3
    assert 0.58 < champs < 0.75
    full = LeagueSlotMapper(4, 4)                            # every slot trainable: identity
    assert np.array_equal(full.assign(["x", "y"]), np.tile(np.arange(4), (2, 1)))


def test_league_actions_route_each_slot_to_its_module():
    import torch
    from gym_continuousdoubleauction_amd.league import league_actions
    m = LeagueSlotMapper(4, 1, 3, champion_weight=2.0)
    m.add_champion()
    idx = m.assign([f"ep{i}" for i in range(50)])
    obs = torch.arange(50, dtype=torch.float32).view(50, 1).repeat(1, 168)

    def const_module(tag):                                   # category = tag, price = the market index it was shown
        def f(o):
            k = o.shape[0]
            return (torch.full((k,), tag, dtype=torch.int32), torch.zeros(k), torch.ones(k), o[:, 0].to(torch.int32), torch.zeros(k, dtype=torch.int32))
        return f
    mods = {name: const_module(i) for i, name in enumerate(m.available_modules)}
    cat, mean, sigma, price, off = league_actions(m, idx, mods, obs)
    assert np.array_equal(cat.numpy(), idx) and (sigma == 1).all() and (mean == 0).all()
    assert np.array_equal(price.numpy(), np.tile(np.arange(50)[:, None], (1, 4)))
