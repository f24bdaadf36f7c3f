"""Dict-shaped facades over the batched HIP env.

CDAEnv              - ONE market with the reference's exact MultiAgentEnv surface.  Drop-in for
                      `continuousDoubleAuctionEnv` (continuousDoubleAuction_env.py:21) where RLlib attaches:
                          tune.register_env(name, lambda cfg: CDAEnv(cfg))          # train/train.py:441-443
CDAVecMultiAgentEnv - N markets behind ONE object: the batched attach point (RLlib's `num_envs_per_env_runner`,
                      train/train.py:509-518).  `step_batch` takes / returns device tensors keyed by agent id - views,
                      nothing is copied; `step` / `reset` speak the list-of-dicts vector protocol (one sub-env per
                      market) and cost ONE device-to-host copy per step for the whole batch.

Same constructor config keys and defaults (config/env_defaults.json:8-27), `agents`, `possible_agents`,
`observation_spaces`, `action_spaces`, `reset(*, seed=None, options=None)` and `step(action_dict)` returning
(obs, rewards, terminateds, truncateds, infos) dicts keyed "agent_i" (+ "__all__").  The env state lives on the GPU;
these classes only marshal dicts <-> tensors.

The reference assigns its per-agent RNG draws - and builds the arrival list it shuffles - in the ITERATION order of the caller's
action dict (exchg/action_helper.py:164-170).  So do these facades: the order travels to the kernel in cda_step's `present`
array (0 = not in the dict, else 1 + position in it), and `LOB_actions` lists the decoded orders in that order too.
"""
import numbers
import os
from itertools import chain
from operator import itemgetter

import numpy as np
import torch

from . import _capi as K
from . import spaces as _spaces
from .vec_env import CDAVecEnv, DEC_DTYPE, ACTION_KEYS

try:  # (ray is absent from the build image; tests/test_host_logic.py runs this branch under the stand-ins of tests/golden/shim)
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base
except Exception:  # noqa: BLE001
    try:  # gymnasium without ray: gymnasium.make() insists on a gymnasium.Env subclass (what RLlib's MultiAgentEnv is, too)
        import gymnasium as _gym

        class _Base(_gym.Env):
            def __init__(self):
                pass
    except Exception:  # noqa: BLE001
        class _Base:  # minimal stand-in so the class still constructs without either
            def __init__(self):
                pass

_G_CAT, _G_MEAN, _G_SIGMA = itemgetter("category"), itemgetter("size_mean"), itemgetter("size_sigma")
_SIDES, _TYPES = ("bid", "ask"), ("market", "limit", "modify", "cancel")


def _host_io_default():
    """CDA_FACADE_HOST_IO=0: the staged path (one pinned H2D copy of the actions, one D2H copy of `packed` per step) instead of kernel I/O on pinned host memory"""
    return os.environ.get("CDA_FACADE_HOST_IO", "1") != "0"


_TERM_NAMES = ("nav_term", "order_penalty", "trade_penalty", "drawdown_penalty", "passive_bonus")
_DEC_FIELDS = ("cash", "cash_on_hold", "position_val", "vwap", "nav", "prev_nav", "max_nav")
_INT_FIELDS = ("net_position", "num_trades", "num_trades_step", "num_passive_fills_step", "order_step_placed", "num_rejected_step")
_ALIAS = {"VWAP": "vwap"}


def _entropy_seed():
    """What gymnasium's `np_random` does on an unseeded first reset: fresh OS entropy (64 bits of it here)."""
    return int(np.random.SeedSequence().entropy) & (2 ** 64 - 1)


class _AccountView:
    """One trader's account (envs/account/account.py:12-53) as exact Decimals, read from and written to the device
    record (`traders[i].acc.cash = Decimal(900)`, as the reference's own tests do, test/test_accounting.py:143-150)."""

    def __init__(self, env, idx):
        object.__setattr__(self, "_env", env)
        object.__setattr__(self, "_idx", idx)

    def __getattr__(self, name):
        f = _ALIAS.get(name, name)
        acc = self._env._vec.get_state(self._env._market).acc[self._idx]
        if f in _DEC_FIELDS:
            return K.dec_to_decimal(getattr(acc, f))
        if f in _INT_FIELDS:
            return int(getattr(acc, f))
        raise AttributeError(name)

    def __setattr__(self, name, value):
        import decimal
        f = _ALIAS.get(name, name)
        vec, mk = self._env._vec, self._env._market
        s = vec.get_state(mk)
        acc = s.acc[self._idx]
        if f in _DEC_FIELDS:
            setattr(acc, f, K.decimal_to_dec(decimal.Decimal(value)))
        elif f in _INT_FIELDS:
            if int(value) != value:
                raise ValueError(f"{name} is an integer count")
            setattr(acc, f, int(value))
        else:
            raise AttributeError(name)
        vec.set_state(mk, s)                             # (the dump holds the history oldest frame first; set_state re-bases the ring at slot 0)


class _TraderView:
    def __init__(self, env, idx):
        self.ID = idx
        self.acc = _AccountView(env, idx)


class _ActionStage:
    """Host staging of one step's actions for n markets: ONE pinned buffer [category | size_mean | size_sigma | price |
    price_offset | present] filled through numpy views and moved with ONE host-to-device copy."""
    _FIELDS = (("category", np.int32), ("size_mean", np.float32), ("size_sigma", np.float32), ("price", np.int32), ("price_offset", np.int32))

    def __init__(self, n, a, device):
        words = n * a
        self.host = torch.zeros(5 * 4 * words + (words + 15) // 16 * 16, dtype=torch.uint8).pin_memory()
        self.dev = torch.zeros_like(self.host, device=device)
        hn = self.host.numpy()
        self.np, self.t = {}, {}
        for j, (k, dt) in enumerate(self._FIELDS):
            self.np[k] = hn[j * 4 * words:(j + 1) * 4 * words].view(dt).reshape(n, a)
            self.t[k] = self.dev[j * 4 * words:(j + 1) * 4 * words].view(torch.int32 if dt is np.int32 else torch.float32).view(n, a)
        self.np["present"] = hn[20 * words:21 * words].reshape(n, a)
        self.t["present"] = self.dev[20 * words:21 * words].view(n, a)
        # the SAME arrays as the kernel sees them when it reads the pinned block in place (CDAVecEnv.step_host_io): pinned host memory is device-addressable at its own address
        base = self.host.data_ptr()
        self.host_ptrs = tuple(base + j * 4 * words for j in range(5)) + (base + 20 * words,)
        self._hn = hn
        self.flat = {k: v.reshape(-1) for k, v in self.np.items()}

    def clear(self):
        self._hn.fill(0)
        self.np["price_offset"].fill(1)                   # neutral 'join' (action_helper.py:256-258)

    def upload(self):
        self.dev.copy_(self.host, non_blocking=True)
        return self.t


def _jsonable(x):
    """model_action as plain Python (info_helper.py:4-20 sanitises the echoed action for json.dumps): numpy scalars and
    arrays become numbers and nested lists, containers are rebuilt around their sanitised members."""
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (np.ndarray, np.generic)):
        return np.asarray(x).tolist()
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    return x


def _nan_none(x):
    return None if x != x else float(x)


_DEC_CTX = None


def _nav_words(a):
    """uint8[..., 16] decimals -> nested lists of their four 32-bit words"""
    return np.ascontiguousarray(a).view(np.uint32).reshape(a.shape[:-1] + (4,)).tolist()


def _dec_str(row):
    """str(Decimal) of one 16-byte decimal as its four little-endian 32-bit words (w0, w1, w2, exp:int16 | sign << 16): the exact sign / coefficient / exponent triple
    (K.dec_to_decimal builds the same object digit by digit; this is the per-step path of the info dicts)."""
    global _DEC_CTX
    import decimal
    if _DEC_CTX is None:
        _DEC_CTX = decimal.Context(prec=60)                       # a 96-bit coefficient has at most 29 digits: scaleb never rounds
    w0, w1, w2, w3 = row
    exp = (w3 & 0xFFFF) - ((w3 & 0x8000) << 1)
    d = decimal.Decimal(w0 | (w1 << 32) | (w2 << 64))
    if w3 & 0xFF0000:
        d = d.copy_negate()
    return str(d.scaleb(exp, _DEC_CTX))


def _info_dict(info, i, k, reward, model_action):
    """info[agent k of market i] as the reference builds it (exchg/info_helper.py:30-116) from the host copy of the SoA info tensors (`info`: a _Snap, or a dict of
    numpy arrays).  Every field goes through ONE .tolist() of its array per step (cached in the _Snap): indexing a numpy array element by element and converting
    each scalar costs five times as much."""
    rows = info.rows if isinstance(info, _Snap) else (lambda name: _nav_words(info[name]) if name == "nav" else info[name].tolist())
    terms = rows("reward_terms")[i][k]
    d = {
        "reward": reward, "NAV": _dec_str(rows("nav")[i][k]), "num_trades": rows("num_trades")[i][k],
        "net_position": rows("net_position")[i][k], "VWAP": rows("vwap")[i][k], "cash": rows("cash")[i][k],
        "cash_on_hold": rows("cash_on_hold")[i][k], "position_val": rows("position_val")[i][k],
        "drawdown": rows("drawdown")[i][k], "max_nav": rows("max_nav")[i][k],
        "num_trades_step": rows("num_trades_step")[i][k],
        "num_passive_fills_step": rows("num_passive_fills_step")[i][k],
        "order_step_placed": rows("order_step_placed")[i][k], "num_rejected_step": rows("num_rejected_step")[i][k],
        "is_pass_action": bool(rows("is_pass_action")[i][k]),
        "reward_terms": dict(zip(_TERM_NAMES, terms)),
        "last_price": rows("last_price")[i], "best_bid": _nan_none(rows("best_bid")[i]), "best_ask": _nan_none(rows("best_ask")[i]),
        "spread": _nan_none(rows("spread")[i]),
    }
    if model_action is not None:
        d["model_action"] = _jsonable(model_action)
    return d


class _Snap:
    """One step's host snapshot of `packed` (a private copy of the pinned block): the info arrays as numpy views, each built on first use - a consumer that reads no
    info dict pays for none of the ~20 views.  Indexed like the info dict of CDAVecEnv.unpack_host."""
    __slots__ = ("h", "lay", "cache", "lists")
    _NP = {torch.uint8: np.uint8, torch.int32: np.int32, torch.float64: np.float64}

    def __init__(self, h, lay):
        self.h, self.lay, self.cache, self.lists = h, lay, {}, {}

    def rows(self, name):
        """the field as nested Python lists ([market][agent]...; "nav": the four 32-bit words of each decimal), converted once per step"""
        v = self.lists.get(name)
        if v is None:
            a = self[name]
            v = self.lists[name] = _nav_words(a) if name == "nav" else a.tolist()
        return v

    def __getitem__(self, name):
        v = self.cache.get(name)
        if v is None:
            o, dt, shape, nbytes = self.lay[name]
            v = self.cache[name] = self.h[o:o + nbytes].view(self._NP[dt]).reshape(shape)
        return v


class _LazyInfo(dict):
    """One agent's info dict of one market, built on first use.  The vector protocol hands out N x A of these per step; a consumer
    that reads none of them (or one field of a few) does not pay for the ~25 conversions each costs.  A real dict once touched:
    every reading method fills it first (json.dumps, ==, iteration, .get(), len() all see the full content).  It reads the
    step's OWN host snapshot (kept alive here), so it stays valid after later steps."""
    __slots__ = ("_src",)

    def __init__(self, info, i, k, reward, model_action):
        # "reward" is stored at once: CPython's C JSON encoder prints "{}" for a dict whose STORAGE is empty without asking it
        # anything; with one real entry it goes through .items(), which fills the rest
        super().__init__(reward=reward)
        self._src = (info, i, k, reward, model_action)

    def _fill(self):
        src = getattr(self, "_src", None)                  # (unset while a pickle is being loaded into a fresh object)
        if src is not None:
            self._src = None
            dict.update(self, _info_dict(*src))

    def _wrap(name):                                      # noqa: N805 - class-body helper
        base = getattr(dict, name)

        def method(self, *a, **kw):
            self._fill()
            return base(self, *a, **kw)
        method.__name__ = name
        return method
    for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "__eq__", "__ne__", "__repr__", "__reduce_ex__", "__or__", "__ror__",
               "get", "items", "keys", "values", "copy", "pop", "setdefault", "update", "__setitem__", "__delitem__", "popitem", "__reversed__"):
        locals()[_n] = _wrap(_n)
    del _n, _wrap
    __hash__ = None


class _DictSurface(_Base):
    """What both facades share: config pickup, agent ids and spaces (continuousDoubleAuction_env.py:27-137)."""
    metadata = {"render.modes": ["human"]}

    def _init_surface(self, vec):
        cfg = vec.config
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
        self._agent_index = {a: i for i, a in enumerate(agent_ids)}
        self.agents = list(agent_ids)
        self.possible_agents = list(agent_ids)
        self._agents_t = tuple(agent_ids)
        self._false_t = dict.fromkeys(agent_ids, False)
        self._present_row = np.arange(1, self.num_of_agents + 1, dtype=np.uint8)
        lay = vec.slab_layout
        self._o_obs, self._o_rew, self._o_term, self._o_trunc = lay["obs"], lay["reward"], lay["terminated"], lay["truncated"]
        obs_space = _spaces.observation_space(self.n_hist)
        self.observation_spaces = {a: obs_space for a in agent_ids}
        act_space = _spaces.action_space()              # ONE shared Dict object, as in the reference
        self.action_spaces = {a: act_space for a in agent_ids}

    def get_action_space(self, agent_id):
        return self.action_spaces[agent_id]

    def get_observation_space(self, agent_id):
        return self.observation_spaces[agent_id]

    def render(self):
        return None

    # ---- dict <-> array marshalling of ONE market ------------------------------------------------------------
    def _encode(self, actions, cat, mean, sigma, price, off, present):
        """One market's action dict -> row views of the [*, A] host arrays (action_helper.py:145-172, :241-283)."""
        A = self.num_of_agents
        index = self._agent_index
        nd = np.ndarray
        for pos, (key, act) in enumerate(actions.items()):
            a = index.get(key)
            if a is None:
                a = int(str(key).split("_")[1])
                if not 0 <= a < A:
                    raise KeyError(key)
            c = act["category"]
            if not 0 <= c <= 8:
                raise KeyError(c)                               # _CATEGORY_MAP lookup (action_helper.py:266)
            present[a] = 1 + pos                                # the dict's iteration order decides who draws which normal (:164-170)
            cat[a] = c
            sm, ss = act["size_mean"], act["size_sigma"]        # Box(shape=(1,)) samples; anything array-like or scalar is accepted
            mean[a] = sm[0] if type(sm) is nd and sm.ndim == 1 else np.asarray(sm, dtype=np.float32).reshape(-1)[0]
            sg = ss[0] if type(ss) is nd and ss.ndim == 1 else np.asarray(ss, dtype=np.float32).reshape(-1)[0]
            if sg < 0:
                raise ValueError("scale < 0")                   # numpy's Generator.normal
            sigma[a] = sg
            price[a] = act.get("price", 0)
            off[a] = act.get("price_offset", 1)                 # neutral 'join' (action_helper.py:256-258)

    def _encode_all(self, action_dicts, st):
        """Every market's action dict -> the staging block `st`.  The common case - every dict lists all agents in their canonical order and every value is a
        size-1 array or numpy scalar - is gathered column-wise (2 us per market instead of 5); anything else, and anything the per-agent statement would refuse
        (category outside 0..8, negative sigma), goes through _encode market by market, which raises what the reference raises where it raises it."""
        agents_t, A = self._agents_t, self.num_of_agents
        try:
            for actions in action_dicts:
                if tuple(actions) != agents_t:
                    raise LookupError
            acts = list(chain.from_iterable(map(dict.values, action_dicts)))
            f = st.flat
            f["category"][:] = list(map(_G_CAT, acts))
            f["size_mean"][:] = [x.item() for x in map(_G_MEAN, acts)]
            f["size_sigma"][:] = [x.item() for x in map(_G_SIGMA, acts)]
            f["price"][:] = [a.get("price", 0) for a in acts]
            f["price_offset"][:] = [a.get("price_offset", 1) for a in acts]     # neutral 'join' (action_helper.py:256-258)
            c, sg = f["category"], f["size_sigma"]
            if c.min() < 0 or c.max() > 8 or sg.min() < 0:
                raise LookupError
            st.np["present"][:] = self._present_row
            return
        except (LookupError, AttributeError, ValueError, TypeError, OverflowError):
            pass
        st.clear()
        buf = st.np
        for i, actions in enumerate(action_dicts):
            self._encode(actions, buf["category"][i], buf["size_mean"][i], buf["size_sigma"][i], buf["price"][i], buf["price_offset"][i], buf["present"][i])

    def _canonical(self, actions):
        """(listed agent ids in the dict's order, {canonical agent id: action}): keys are mapped the way _encode resolves them (a key like "agent_02" or "trader_2"
        addresses agent_2: its order and its model_action must not disappear from the outputs)"""
        if actions is None:
            return self.agents, None
        if tuple(actions) == self._agents_t:
            return self.agents, actions
        agents, index, canon = self.agents, self._agent_index, {}
        for key in actions:
            k = index.get(key)
            if k is None:
                k = int(str(key).split("_")[1])
            canon[agents[k]] = key
        return list(canon), {a: actions[key] for a, key in canon.items()}

    def _decode_snap(self, actions, h):
        """ONE market's step from its own host snapshot `h` of the pinned block (CDAEnv's host-I/O path): the reference's five dicts with lazily built info dicts,
        LOB_actions, the pass agents and the bankrupt agents (NAV <= 0: sign set or zero coefficient of the 16-byte decimal)."""
        agents, A = self.agents, self.num_of_agents
        ob = h[self._o_obs:self._o_obs + self.n_hist * 42 * 4].view(np.float32)
        rew = h[self._o_rew:self._o_rew + A * 8].view(np.float64).tolist()
        snap = _Snap(h, self._info_lay)
        terminateds, truncateds = self._false_t.copy(), self._false_t.copy()
        terminateds["__all__"], truncateds["__all__"] = bool(h[self._o_term]), bool(h[self._o_trunc])
        listed, canon = self._canonical(actions)
        lob = snap["lob_actions"][0].tolist()
        index = self._agent_index
        lob_actions = [{"ID": a, "side": _SIDES[row[0]], "type": _TYPES[row[1]], "size": row[2], "price": float(row[3])}
                       for a, row in ((a, lob[index[a]]) for a in listed) if row[0] >= 0]
        pass_agents = {a for a, p in zip(agents, snap["is_pass_action"][0].tolist()) if p}
        nv = snap["nav"][0]
        bankrupt = {a for a, b in zip(agents, ((nv[:, 14] != 0) | ~nv[:, :12].any(axis=1)).tolist()) if b}
        get = canon.get if canon is not None else (lambda a: None)
        infos = {a: _LazyInfo(snap, 0, k, rew[k], get(a)) for k, a in enumerate(agents)}
        return (dict.fromkeys(agents, ob), dict(zip(agents, rew)), terminateds, truncateds, infos), lob_actions, pass_agents, bankrupt

    def _decode(self, i, actions, obs, rew, term, trunc, info):
        """Market i's rows of the host copy -> the reference's five dicts (info_helper.py:30-116)."""
        agents = self.agents
        ob = obs[i]
        nav = info["nav"][i].view(DEC_DTYPE).reshape(self.num_of_agents)
        observations = {a: ob for a in agents}              # the SAME array object for every agent
        rewards = {a: float(rew[i, k]) for k, a in enumerate(agents)}
        terminateds = {a: False for a in agents}
        truncateds = {a: False for a in agents}
        terminateds["__all__"] = bool(term[i])
        truncateds["__all__"] = bool(trunc[i])
        lob = info["lob_actions"][i]
        # env.LOB_actions (continuousDoubleAuction_env.py:284-285): the decoded orders in the order of the action dict, passes left out.
        # Keys are mapped to the canonical agent ids the way _encode resolves them (a key like "agent_02" or "trader_2" addresses agent_2:
        # its order and its model_action must not disappear from the outputs)
        index = {a: k for k, a in enumerate(agents)}
        if actions is not None:
            canon = {}
            for key in actions:
                k = index.get(key)
                if k is None:
                    k = int(str(key).split("_")[1])
                canon[agents[k]] = key
            listed = list(canon)
            actions = {a: actions[key] for a, key in canon.items()}
        else:
            listed = agents
        lob_actions = [
            {"ID": a, "side": ("bid", "ask")[int(row[0])], "type": ("market", "limit", "modify", "cancel")[int(row[1])],
             "size": int(row[2]), "price": float(row[3])}
            for a, row in ((a, lob[index[a]]) for a in listed) if row[0] >= 0]
        infos, pass_agents, bankrupt = {}, set(), set()
        for k, a in enumerate(agents):
            if K.dec_to_decimal(nav[k]) <= 0:
                bankrupt.add(a)
            if info["is_pass_action"][i, k]:
                pass_agents.add(a)
            infos[a] = _info_dict(info, i, k, rewards[a], actions.get(a) if actions is not None else None)
        return (observations, rewards, terminateds, truncateds, infos), lob_actions, pass_agents, bankrupt


class CDAEnv(_DictSurface):
    def __init__(self, config=None, device="cuda:0"):
        super().__init__()
        self.config = dict(config or {})
        self._vec = CDAVecEnv(self.config, n_markets=1, device=device, with_info=True)
        self._market = 0
        self._init_surface(self._vec)
        self.traders = [_TraderView(self, i) for i in range(self.num_of_agents)]
        self.done_set = set()
        self.LOB_actions = None
        self.pass_agents = set()
        self.t_step = 0
        self.model_actions = None
        self._seeded = False
        A = self.num_of_agents
        self._stage = _ActionStage(1, A, self._vec.device)
        self._info_lay = self._vec.info_layout
        # Host-resident step I/O (default): the step kernel reads the staged actions from, and writes obs | reward | flags | info into, pinned host memory - a step is
        # one launch + one stream synchronisation.  CDA_FACADE_HOST_IO=0: ONE pinned buffer receives a whole step in one D2H copy, the actions go up in one H2D copy.
        self._host_io = _host_io_default()
        if self._host_io:
            self._hio = self._vec.bind_host_io()
        else:
            self._host = torch.empty(self._vec.packed.numel(), dtype=torch.uint8).pin_memory()

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
        conv = lambda o: {"price": int(o[0]), "quantity": int(o[1]), "trade_id": int(o[2]), "order_id": int(o[3]),  # noqa: E731
                          "timestamp": int(o[4])}
        bids, asks = self._vec.get_book(0)              # the WHOLE book (cda_get_book), not the first CDA_BOOK_CAP_MAX orders of the state dump
        return [conv(o) for o in bids], [conv(o) for o in asks]

    # -- reset / step ---------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        """reset(seed=s): seeded episode.  reset(seed=None): the FIRST unseeded reset of an env draws a seed from OS entropy
        (gymnasium's np_random does; several EnvRunners each constructing this env must not replay one stream), later
        unseeded resets continue the stream (continuousDoubleAuction_env.py:186-188)."""
        if seed is None and not self._seeded:
            seed = _entropy_seed()
        self._seeded = True
        seeds = None if seed is None else np.array([int(seed)], dtype=np.uint64)
        if self._host_io:
            self._vec.reset_host_io(seed=seeds)
            self._vec.sync_host_io()
            ob = self._hio[self._o_obs:self._o_obs + self.n_hist * 42 * 4].view(np.float32).copy()
        else:
            ob = self._vec.reset(seed=seeds)[0].cpu().numpy()
        self.done_set = set()
        self.LOB_actions = None
        self.pass_agents = set()
        self.t_step = 0
        observations = {a: ob for a in self.agents}        # the SAME array object for every agent
        infos = {a: {} for a in self._agent_ids}
        return observations, infos

    def step(self, actions):
        st = self._stage
        vec = self._vec
        if self._host_io:
            self._encode_all((actions,), st)
            self.model_actions = actions
            vec.step_host_io(st.host_ptrs)                     # one launch: actions read from, outputs written to, pinned host memory
            vec.sync_host_io()
            # this step's own snapshot (a 2-KB memcpy): the observation row and the lazily built info dicts keep reading it after the block has moved on
            out, self.LOB_actions, self.pass_agents, bankrupt = self._decode_snap(actions, self._hio.copy())
            self.done_set |= bankrupt
            self.t_step += 1
            return out
        st.clear()
        buf = st.np
        self._encode(actions, buf["category"][0], buf["size_mean"][0], buf["size_sigma"][0], buf["price"][0], buf["price_offset"][0], buf["present"][0])
        self.model_actions = actions
        t = st.upload()                                        # ONE host-to-device copy of the step's actions
        vec.step(t["category"], t["size_mean"], t["size_sigma"], t["price"], t["price_offset"], t["present"])
        self._host.copy_(vec.packed)                           # the ONE device-to-host copy of the step (synchronous)
        obs, rew, term, trunc, info = vec.unpack_host(self._host)
        out, self.LOB_actions, self.pass_agents, bankrupt = self._decode(0, actions, obs.copy(), rew, term, trunc, info)
        self.done_set |= bankrupt
        self.t_step += 1
        return out

    def close(self):
        self._vec.close()


class CDAVecMultiAgentEnv(_DictSurface):
    """N markets as N sub-envs of one object (the batched attach point).

    Tensor API (no host round trip, everything is a VIEW of the env's output buffers):
        obs, infos = env.reset_batch(seed=None)
        obs, rewards, terminateds, truncateds, infos = env.step_batch(actions)
      `actions`: {key: tensor [N, A]} for the five action keys (+ optional "present"), or {agent_id: {key: tensor [N]}}.
      obs[agent] f32[N, n_hist*42] (the same tensor for every agent, as the reference hands every agent one array),
      rewards[agent] f64[N] (column a of the [N, A] reward tensor), terminateds / truncateds [agent] all-False bool[N]
      plus "__all__" bool[N], infos[agent][field] [N, ...] (column a of the SoA info tensors; "nav" as 16-byte decimal
      triples, see CDAVecEnv.nav_decimals).
    Vector protocol (lists of per-market dicts, the shape RLlib's vector env runners consume):
        obs_list, info_list = env.reset(seed=None)
        obs_list, rew_list, term_list, trunc_list, info_list = env.step(list_of_action_dicts)
      one device-to-host copy per step for the whole batch; each market's dicts are what CDAEnv returns for it.
    With config["auto_reset"] a finished market restarts in place (see include/cda.h); otherwise reset(mask=...).
    """

    def __init__(self, config=None, num_envs=1, device="cuda:0", with_info=True, groups=1):
        super().__init__()
        self.config = dict(config or {})
        self.num_envs = int(num_envs)
        self._vec = CDAVecEnv(self.config, n_markets=self.num_envs, device=device, with_info=with_info, groups=groups)
        self._init_surface(self._vec)
        self._seeded = False
        N, A = self.num_envs, self.num_of_agents
        self._false = torch.zeros(N, dtype=torch.bool, device=self._vec.device)
        self._host = None
        self._stage = None
        self._info_lay = self._vec.info_layout
        self._host_io = _host_io_default() and int(groups) == 1 and bool(with_info)    # (see CDAEnv: kernel I/O on pinned host memory; single-launch envs)
        self._hio = None

    @property
    def vec(self):
        """The underlying CDAVecEnv (flags(), nav_conservation(), get_state(i), ...)."""
        return self._vec

    # ---- tensor API -------------------------------------------------------------------------------------------
    def _views(self, obs, rew, term, trunc, info):
        agents = self.agents
        observations = {a: obs for a in agents}
        rewards = {a: rew[:, k] for k, a in enumerate(agents)}
        terminateds = {a: self._false for a in agents}
        truncateds = {a: self._false for a in agents}
        terminateds["__all__"], truncateds["__all__"] = term, trunc
        infos = {a: {} for a in agents}
        for name, t in info.items():
            per_agent = t.dim() >= 2 and t.shape[1] == self.num_of_agents and name not in ("last_price", "best_bid", "best_ask", "spread")
            for k, a in enumerate(agents):
                infos[a][name] = t[:, k] if per_agent else t
        return observations, rewards, terminateds, truncateds, infos

    def _seed_arg(self, seed):
        if seed is None and not self._seeded:
            seed = _entropy_seed()
        self._seeded = True
        if isinstance(seed, numbers.Integral):                      # market i gets SeedSequence(seed + i)
            return (np.uint64(int(seed) & (2 ** 64 - 1)) + np.arange(self.num_envs, dtype=np.uint64)).astype(np.uint64)
        return seed

    def reset_batch(self, seed=None, mask=None):
        obs = self._vec.reset(seed=self._seed_arg(seed), mask=mask)
        return {a: obs for a in self.agents}, {a: {} for a in self.agents}

    def step_batch(self, actions):
        first = next(iter(actions.values()))
        if isinstance(first, dict):                                 # {agent: {key: [N]}} -> {key: [N, A]}
            # `present` carries the dict's iteration order (0 = absent, else 1 + position): needed when agents are missing or the
            # keys do not come in ascending agent order
            order = [a for a in actions if a in self._agent_ids]
            present = None
            if order != self.agents:
                present = torch.zeros((self.num_envs, self.num_of_agents), dtype=torch.uint8, device=self._vec.device)
            cols = {k: [] for k in ACTION_KEYS}
            dts = {"category": torch.int32, "size_mean": torch.float32, "size_sigma": torch.float32, "price": torch.int32, "price_offset": torch.int32}
            dev, n = self._vec.device, self.num_envs
            for j, a in enumerate(self.agents):
                act = actions.get(a)
                if act is not None and present is not None:
                    present[:, j] = 1 + order.index(a)
                for k in ACTION_KEYS:
                    if act is None:
                        cols[k].append(torch.zeros(n, dtype=dts[k], device=dev))
                    else:
                        v = act.get(k, 1 if k == "price_offset" else 0) if k in ("price", "price_offset") else act[k]
                        v = torch.as_tensor(v, device=dev).to(dts[k]).reshape(-1)
                        cols[k].append(v.expand(n) if v.numel() == 1 else v.reshape(n))
            actions = {k: torch.stack(cols[k], dim=1) for k in ACTION_KEYS}
            if present is not None:
                actions["present"] = present
        obs, rew, term, trunc, info = self._vec.step(actions)      # (groups > 1: forked from and joined back into the caller's stream)
        return self._views(obs, rew, term, trunc, info)

    # ---- vector protocol: one sub-env per market --------------------------------------------------------------
    def reset(self, *, seed=None, options=None, mask=None):
        obs = self._vec.reset(seed=self._seed_arg(seed), mask=mask).cpu().numpy()
        return [{a: obs[i] for a in self.agents} for i in range(self.num_envs)], [{a: {} for a in self.agents} for _ in range(self.num_envs)]

    def step(self, action_dicts):
        if len(action_dicts) != self.num_envs:
            raise ValueError(f"need {self.num_envs} action dicts, one per market")
        if not self._vec.with_info:
            raise ValueError("the vector protocol builds info dicts: construct with with_info=True")
        vec = self._vec
        if self._stage is None:
            self._stage = _ActionStage(self.num_envs, self.num_of_agents, vec.device)
            if self._host_io:
                self._hio = vec.bind_host_io()
            else:
                self._host = torch.empty(vec.packed.numel(), dtype=torch.uint8).pin_memory()
        st = self._stage
        self._encode_all(action_dicts, st)
        if self._host_io:
            vec.step_host_io(st.host_ptrs)                         # one launch: actions read from, outputs written to, pinned host memory
            vec.sync_host_io()
            h = self._hio.copy()                                   # this step's own snapshot (1.2 KB per market)
            n, A, od = self.num_envs, self.num_of_agents, self.n_hist * 42
            obs = h[self._o_obs:self._o_obs + n * od * 4].view(np.float32).reshape(n, od)
            rew = h[self._o_rew:self._o_rew + n * A * 8].view(np.float64).reshape(n, A)
            term, trunc = h[self._o_term:self._o_term + n], h[self._o_trunc:self._o_trunc + n]
            info = _Snap(h, self._info_lay)
        else:
            t = st.upload()                                        # ONE host-to-device copy of the batch's actions
            vec.step(t["category"], t["size_mean"], t["size_sigma"], t["price"], t["price_offset"], t["present"])
            self._host.copy_(vec.packed)                           # ONE device-to-host copy for the whole batch
            # this step's own snapshot: the observations rows and the lazily built info dicts keep reading it after the staging buffer
            # has moved on (one 1.2-KB-per-market memcpy instead of N x A x 25 eager conversions)
            hc = self._host.numpy().copy()
            obs, rew, term, trunc, _ = vec.unpack_host(hc)
            info = _Snap(hc, self._info_lay)
        agents = self.agents
        rew_l, term_l, trunc_l = rew.tolist(), term.tolist(), trunc.tolist()
        false_t = dict.fromkeys(agents, False)
        obs_out, rew_out, term_out, trunc_out, info_out = [], [], [], [], []
        for i, actions in enumerate(action_dicts):
            ob, r = obs[i], rew_l[i]
            obs_out.append(dict.fromkeys(agents, ob))              # the SAME array object for every agent
            rew_out.append(dict(zip(agents, r)))
            te = false_t.copy(); te["__all__"] = bool(term_l[i])
            tr = false_t.copy(); tr["__all__"] = bool(trunc_l[i])
            term_out.append(te); trunc_out.append(tr)
            info_out.append({a: _LazyInfo(info, i, k, r[k], actions.get(a)) for k, a in enumerate(agents)})
        return obs_out, rew_out, term_out, trunc_out, info_out

    def close(self):
        self._vec.close()
