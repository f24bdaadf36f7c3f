"""Random-agent driver - the build's counterpart of the reference's `CDA_rand.run_random`
(gym_continuousDoubleAuction/CDA_rand.py:40-85): build the env, reset(seed), sample every agent's action
from its action space each step, step to the horizon or until an `__all__` flag is raised.

    python -m gym_continuousdoubleauction_amd.cda_rand --agents 4 --steps 1000 --seed 123
"""
import argparse
import sys

# config/cli_defaults.json:9-14 of the reference
CLI_DEFAULTS = {"num_agents": 4, "max_step": 1000, "init_cash": 1000000, "is_render": False, "seed": None}


def run_random(num_agents=None, max_step=None, init_cash=None, is_render=None, seed=None, device="cuda:0"):
    """Returns the number of steps actually taken (same contract as the reference)."""
    from .env import CDAEnv
    num_agents = CLI_DEFAULTS["num_agents"] if num_agents is None else num_agents
    max_step = CLI_DEFAULTS["max_step"] if max_step is None else max_step
    init_cash = CLI_DEFAULTS["init_cash"] if init_cash is None else init_cash
    is_render = CLI_DEFAULTS["is_render"] if is_render is None else is_render
    env = CDAEnv({"num_of_agents": num_agents, "init_cash": init_cash, "max_step": max_step, "is_render": is_render},
                 device=device)
    env.reset(seed=seed)
    if seed is not None:
        for agent_id in env.agents:
            env.action_spaces[agent_id].seed(seed)
    steps = 0
    for _ in range(max_step):
        actions = {agent_id: env.action_spaces[agent_id].sample() for agent_id in env.agents}
        _obs, _rewards, terminateds, truncateds, _infos = env.step(actions)
        steps += 1
        if terminateds.get("__all__", False) or truncateds.get("__all__", False):
            break
    env.close()
    return steps


def main(argv=None):
    p = argparse.ArgumentParser(description="Random-agent CDA simulation on one MI355X.")
    p.add_argument("--agents", type=int, default=CLI_DEFAULTS["num_agents"])
    p.add_argument("--steps", type=int, default=CLI_DEFAULTS["max_step"])
    p.add_argument("--init-cash", type=int, default=CLI_DEFAULTS["init_cash"])
    p.add_argument("--seed", type=int, default=CLI_DEFAULTS["seed"])
    args = p.parse_args(argv)
    steps = run_random(args.agents, args.steps, args.init_cash, False, args.seed)
    print(f"completed {steps} steps with {args.agents} random agents.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
