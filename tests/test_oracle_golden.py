"""Pins the CPU oracle (oracle/cda_oracle.c) against golden vectors cut from the reference itself
(tests/golden/make_goldens.py, run in the build container with the reference imported).
Every recorded field is compared bit for bit: obs (f32 bits), rewards (f64 bits), Decimal triples,
book in queue order, LOB clocks, numpy PCG64 state, decoded orders and execution order."""
import json

import pytest

import golden_util as G
import oracle_lib as O

GROUPS = G.group_by_config(G.trace_names())


@pytest.mark.parametrize("key", sorted(GROUPS), ids=lambda k: "+".join(r["name"] for r in GROUPS[k])[:60])
def test_oracle_matches_reference_goldens(key):
    recs = GROUPS[key]
    env = O.OracleEnv(json.loads(key), n_markets=len(recs))
    steps = G.run_group(env, recs, state_every=1, trace_getter=lambda: env.trace)
    assert steps > 0
    env.close()


@pytest.mark.parametrize("name", ["perm_s91", "perm8_s92"])
def test_dict_key_order_is_a_pure_relabelling_of_agents(name):
    """The reference hands its RNG draws out in the ITERATION order of the action dict (action_helper.py:145-172); the
    build always works in ascending agent order.  For traces the reference produced from dicts in a fixed non-ascending
    key order: replayed as recorded they must NOT match (the order matters), renamed so that agent k is the k-th key
    the reference iterated they match bit for bit (it is only a renaming)."""
    raw, ren = G.load(name, relabel=False), G.load(name)
    assert list(raw["dict_order"]) != sorted(raw["dict_order"])
    env = O.OracleEnv(ren["config"], n_markets=1)
    assert G.run_group(env, [ren], state_every=1, trace_getter=lambda: env.trace) == raw["cat"].shape[0]
    env.close()
    env = O.OracleEnv(raw["config"], n_markets=1)
    with pytest.raises(AssertionError):
        G.run_group(env, [raw], state_every=1)
    env.close()
