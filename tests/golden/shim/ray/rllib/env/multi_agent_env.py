"""Stand-in for ray.rllib.env.multi_agent_env (container-only; see gymnasium shim)."""
import gymnasium as gym


class MultiAgentEnv(gym.Env):
    def __init__(self):
        pass
