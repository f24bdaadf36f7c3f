"""CDAEnv - the single-market facade with the reference's exact dict-shaped surface.

Drop-in for `continuousDoubleAuctionEnv` (continuousDoubleAuction_env.py:21) where RLlib attaches:
    tune.register_env(name, lambda cfg: CDAEnv(cfg))          # train/train.py:441-443
Same constructor config keys and defaults (config/env_defaults.json:8-27), `agents`,
`possible_agents`, `observation_spaces`, `action_spaces`, `reset(*, seed=None, options=None)` and
`step(action_dict)` returning (obs, rewards, terminateds, truncateds, infos) dicts keyed "agent_i"
(+ "__all__").  The env state lives on the GPU; this class only marshals dicts <-> tensors.

One deliberate canonicalisation: the reference assigns its per-agent RNG draws in the
caller's dict iteration order; here agents are always processed in ascending index order.
"""
import numpy as np

from . import _capi as K
from . import spaces as _spaces
from .vec_env import CDAVecEnv, DEC_DTYPE

try:  # pragma: no cover - ray is absent from the build image
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base
except Exception:  # noqa: BLE001
    class _Base:  # minimal stand-in so the class still constructs without RLlib
        def __init__(self):
            pass

_TERM_NAMES = ("nav_term", "order_penalty", "trade_penalty", "drawdown_penalty", "passive_bonus")


class _AccountView:
    """Read-only view of one trader's account (envs/account/account.py:12-53) as exact Decimals."""

    def __init__(self, env, idx):
        self._env, self._idx = env, idx

    def _acc(self):
        return self._env._vec.get_state(0).acc[self._idx]

    def __getattr__(self, name):
        alias = {"VWAP": "vwap"}
        f = alias.get(name, name)
        acc = self._acc()
        if f in ("cash", "cash_on_hold", "position_val", "vwap", "nav", "prev_nav", "max_nav"):
            return K.dec_to_decimal(getattr(acc, f))
        if f in ("net_position", "num_trades", "num_trades_step", "num_passive_fills_step", "order_step_placed",
                 "num_rejected_step"):
            return int(getattr(acc, f))
        raise AttributeError(name)


class _TraderView:
    def __init__(self, env, idx):
        self.ID = idx
        self.acc = _AccountView(env, idx)


class CDAEnv(_Base):
    metadata = {"render.modes": ["human"]}

    def __init__(self, config=None, device="cuda:0"):
        super().__init__()
        self.config = dict(config or {})
        self._vec = CDAVecEnv(self.config, n_markets=1, device=device, with_info=True)
        cfg = self._vec.config
        self.num_of_agents = int(cfg["num_of_agents"])
        self.max_step = int(cfg["max_step"])
        self.n_hist = int(cfg["n_hist"])
        self.min_tick = cfg["tick_size"]
        self.tick_size = cfg["tick_size"]
        self.is_render = cfg["is_render"]
        self.init_cash = cfg["init_cash"]
        self.k_rows, self.book_rows, self.extra_dim = 10, 4, 2
        self.book_dim = 40
        self.snapshot_dim = 42
        agent_ids = [f"agent_{i}" for i in range(self.num_of_agents)]
        self._agent_ids = set(agent_ids)
        self.agents = list(agent_ids)
        self.possible_agents = list(agent_ids)
        obs_space = _spaces.observation_space(self.n_hist)
        self.observation_spaces = {a: obs_space for a in agent_ids}
        act_space = _spaces.action_space()              # ONE shared Dict object, as in the reference
        self.action_spaces = {a: act_space for a in agent_ids}
        self.traders = [_TraderView(self, i) for i in range(self.num_of_agents)]
        self.done_set = set()
        self.LOB_actions = None
        self.pass_agents = set()
        self.t_step = 0
        self.model_actions = None

    # -- spaces -------------------------------------------------------------------------
    def get_action_space(self, agent_id):
        return self.action_spaces[agent_id]

    def get_observation_space(self, agent_id):
        return self.observation_spaces[agent_id]

    # -- diagnostics some reference tests read --------------------------------------------
    @property
    def last_price(self):
        return float(self._vec.get_state(0).last_price)

    @last_price.setter
    def last_price(self, value):
        if float(value) != int(value):
            raise ValueError("last_price must be integer valued at tick_size 1")
        s = self._vec.get_state(0)
        s.last_price = int(value)
        self._vec.set_state(0, s)

    @property
    def agg_LOB_raw(self):
        return self._vec.raw_snapshot()[0].cpu().numpy()

    def book(self):
        """(bids, asks): lists of dicts in queue order (best price first, FIFO inside a level)."""
        s = self._vec.get_state(0)
        conv = lambda o: {"price": o.price, "quantity": o.qty, "trade_id": o.owner, "order_id": o.order_id,  # noqa: E731
                          "timestamp": o.timestamp}
        return [conv(o) for o in s.bids[: s.n_bids]], [conv(o) for o in s.asks[: s.n_asks]]

    # -- reset / step ---------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        obs = self._vec.reset(seed=None if seed is None else np.array([int(seed)], dtype=np.uint64))
        self.done_set = set()
        self.LOB_actions = None
        self.pass_agents = set()
        self.t_step = 0
        ob = obs[0].cpu().numpy()
        observations = {a: ob for a in self.agents}        # the SAME array object for every agent
        infos = {a: {} for a in self._agent_ids}
        return observations, infos

    def step(self, actions):
        A = self.num_of_agents
        cat = np.zeros(A, np.int32); mean = np.zeros(A, np.float32); sigma = np.zeros(A, np.float32)
        price = np.zeros(A, np.int32); off = np.ones(A, np.int32); present = np.zeros(A, np.uint8)
        for key, act in actions.items():
            a = int(str(key).split("_")[1])
            if not 0 <= a < A:
                raise KeyError(key)
            present[a] = 1
            cat[a] = int(act["category"])
            if not 0 <= cat[a] <= 8:
                raise KeyError(int(act["category"]))           # _CATEGORY_MAP lookup (action_helper.py:266)
            mean[a] = np.float32(np.asarray(act["size_mean"], dtype=np.float32).reshape(-1)[0])
            sigma[a] = np.float32(np.asarray(act["size_sigma"], dtype=np.float32).reshape(-1)[0])
            if sigma[a] < 0:
                raise ValueError("scale < 0")                   # numpy's Generator.normal
            price[a] = int(act.get("price", 0))
            off[a] = int(act.get("price_offset", 1))            # neutral 'join' (action_helper.py:256-258)
        self.model_actions = actions
        obs_t, rew_t, term_t, trunc_t, info_t = self._vec.step(cat[None], mean[None], sigma[None], price[None], off[None],
                                                               present[None])
        ob = obs_t[0].cpu().numpy()
        rew = rew_t[0].cpu().numpy()
        info = {k: v[0].cpu().numpy() for k, v in info_t.items()}
        nav = info["nav"].view(DEC_DTYPE).reshape(A)
        obs = {a: ob for a in self.agents}
        rewards = {a: float(rew[i]) for i, a in enumerate(self.agents)}
        terminateds = {a: False for a in self.agents}
        truncateds = {a: False for a in self.agents}
        terminateds["__all__"] = bool(term_t[0].item())
        truncateds["__all__"] = bool(trunc_t[0].item())
        nn = lambda x: None if np.isnan(x) else float(x)       # noqa: E731
        infos = {}
        self.pass_agents = set()
        # env.LOB_actions (continuousDoubleAuction_env.py:284-285): the decoded orders in agent order, passes left out
        self.LOB_actions = [
            {"ID": a, "side": ("bid", "ask")[int(row[0])], "type": ("market", "limit", "modify", "cancel")[int(row[1])],
             "size": int(row[2]), "price": float(row[3])}
            for a, row in zip(self.agents, info["lob_actions"]) if row[0] >= 0]
        for i, a in enumerate(self.agents):
            nav_dec = K.dec_to_decimal(nav[i])
            if nav_dec <= 0:
                self.done_set.add(a)
            if info["is_pass_action"][i]:
                self.pass_agents.add(a)
            d = {
                "reward": rewards[a], "NAV": str(nav_dec), "num_trades": int(info["num_trades"][i]),
                "net_position": int(info["net_position"][i]), "VWAP": float(info["vwap"][i]), "cash": float(info["cash"][i]),
                "cash_on_hold": float(info["cash_on_hold"][i]), "position_val": float(info["position_val"][i]),
                "drawdown": float(info["drawdown"][i]), "max_nav": float(info["max_nav"][i]),
                "num_trades_step": int(info["num_trades_step"][i]),
                "num_passive_fills_step": int(info["num_passive_fills_step"][i]),
                "order_step_placed": int(info["order_step_placed"][i]), "num_rejected_step": int(info["num_rejected_step"][i]),
                "is_pass_action": bool(info["is_pass_action"][i]),
                "reward_terms": {n: float(info["reward_terms"][i, j]) for j, n in enumerate(_TERM_NAMES)},
                "last_price": float(info["last_price"]), "best_bid": nn(info["best_bid"]), "best_ask": nn(info["best_ask"]),
                "spread": nn(info["spread"]),
            }
            if a in actions:
                d["model_action"] = _plain(actions[a])
            infos[a] = d
        self.t_step += 1
        return obs, rewards, terminateds, truncateds, infos

    def render(self):
        return None

    def close(self):
        self._vec.close()


def _plain(value):
    if isinstance(value, np.ndarray):
        return [_plain(v) for v in value.tolist()]
    if isinstance(value, np.generic):
        return value.item()
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    return value
