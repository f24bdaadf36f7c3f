"""The reference's own unit-test KATs (restated as data in kat_scenarios.py) against the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
from kat_scenarios import SCENARIOS, run_scenario


@pytest.mark.parametrize("sc", SCENARIOS, ids=lambda s: s["name"])
def test_oracle_kat(sc):
    env = O.OracleEnv({"num_of_agents": 4, "init_cash": sc["cash"], "max_step": 64, "is_render": False}, n_markets=1)
    env.reset(seeds=np.array([1], np.uint64))
    run_scenario(env, sc)
    env.close()
