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


@pytest.mark.parametrize("name", ["perm_s91", "perm8_s92", "permshuf_s93", "permshuf8_s94"])
def test_dict_key_order_is_honoured(name):
    """The reference hands its RNG draws out - and builds the arrival list it shuffles - in the ITERATION order of the action
    dict (action_helper.py:164-170).  These traces were cut with dicts in a fixed non-ascending key order (perm*) or in a new
    random order every step (permshuf*); `present` carries the order (0 = absent, else 1 + position in the dict).  They replay
    bit for bit AS RECORDED (the parametrised test above does that too); with the order thrown away (a plain 0 / 1 mask) they
    must NOT - the order matters; and a fixed order is a pure renaming of identical traders: restated with agent k = the k-th
    key, in ascending order, the episode matches again."""
    rec = G.load(name)
    env = O.OracleEnv(rec["config"], n_markets=1)
    assert G.run_group(env, [rec], state_every=1, trace_getter=lambda: env.trace) == rec["cat"].shape[0]
    env.close()
    flat = dict(rec, present=(rec["present"] != 0).astype(rec["present"].dtype))
    env = O.OracleEnv(rec["config"], n_markets=1)
    with pytest.raises(AssertionError):
        G.run_group(env, [flat], state_every=1)
    env.close()
    if "dict_order" in rec:
        ren = G.load(name, relabel=True)
        env = O.OracleEnv(ren["config"], n_markets=1)
        assert G.run_group(env, [ren], state_every=1, trace_getter=lambda: env.trace) == rec["cat"].shape[0]
        env.close()
