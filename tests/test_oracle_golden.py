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
