"""GPU: the reference's own unit-test KATs (kat_scenarios.py) against the HIP path via cda_place_order,
and lock-step agreement with the oracle's full state dump after every op."""
import numpy as np
import pytest

from kat_scenarios import SCENARIOS, run_scenario

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sc", SCENARIOS, ids=lambda s: s["name"])
def test_hip_kat(sc):
    from hip_env import HipEnv
    import oracle_lib as O
    cfg = {"num_of_agents": 4, "init_cash": sc["cash"], "max_step": 64, "is_render": False}
    env = HipEnv(cfg, n_markets=1)
    env.reset(seeds=np.array([1], np.uint64))
    run_scenario(env, sc)
    ora = O.OracleEnv(cfg, n_markets=1)
    ora.reset(seeds=np.array([1], np.uint64))
    run_scenario(ora, sc)
    assert bytes(env.get_state(0)) == bytes(ora.get_state(0))
    env.close()
    ora.close()
