"""GPU parity, anchored on the reference: the HIP path (through the C-ABI) replays the golden traces
cut from the real reference and must reproduce every recorded field bit for bit."""
import json

import pytest

import golden_util as G

pytestmark = pytest.mark.gpu

GROUPS = G.group_by_config(G.trace_names())


@pytest.mark.parametrize("key", sorted(GROUPS), ids=lambda k: "+".join(r["name"] for r in GROUPS[k])[:60])
def test_hip_matches_reference_goldens(key):
    from hip_env import HipEnv
    recs = GROUPS[key]
    env = HipEnv(json.loads(key), n_markets=len(recs))
    # full state dump every 8th step (each dump is a synchronous D2H copy), outputs every step
    steps = G.run_group(env, recs, state_every=8)
    assert steps > 0
    assert (env.flags() == 0).all()
    env.close()


@pytest.mark.parametrize("name", ["perm_s91", "perm8_s92"])
def test_dict_key_order_is_a_pure_relabelling_of_agents(name):
    """See tests/test_oracle_golden.py: reference traces cut from action dicts in a fixed NON-ascending key order are
    reproduced by the HIP path once agent k is read as the k-th key the reference iterated - and not otherwise."""
    from hip_env import HipEnv
    raw, ren = G.load(name, relabel=False), G.load(name)
    env = HipEnv(ren["config"], n_markets=1)
    assert G.run_group(env, [ren], state_every=4) == raw["cat"].shape[0]
    assert (env.flags() == 0).all()
    env.close()
    env = HipEnv(raw["config"], n_markets=1)
    with pytest.raises(AssertionError):
        G.run_group(env, [raw], state_every=0)
    env.close()
