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


@pytest.mark.parametrize("name", ["perm_s91", "perm8_s92", "permshuf_s93", "permshuf8_s94"])
def test_dict_key_order_is_honoured(name):
    """See tests/test_oracle_golden.py: reference traces cut from action dicts in a fixed NON-ascending key order, or in a new
    order every step, are reproduced by the HIP path as recorded (`present` carries the order) - and not with the order thrown
    away; a fixed order restated as a renaming of the agents matches as well."""
    from hip_env import HipEnv
    rec = G.load(name)
    env = HipEnv(rec["config"], n_markets=1)
    assert G.run_group(env, [rec], state_every=4) == rec["cat"].shape[0]
    assert (env.flags() == 0).all()
    env.close()
    flat = dict(rec, present=(rec["present"] != 0).astype(rec["present"].dtype))
    env = HipEnv(rec["config"], n_markets=1)
    with pytest.raises(AssertionError):
        G.run_group(env, [flat], state_every=0)
    env.close()
    if "dict_order" in rec:
        ren = G.load(name, relabel=True)
        env = HipEnv(ren["config"], n_markets=1)
        assert G.run_group(env, [ren], state_every=4) == rec["cat"].shape[0]
        env.close()
