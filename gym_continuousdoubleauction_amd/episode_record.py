"""Per-step episode record of the batched env: one Parquet row per (episode, step, agent) in the on-disk
format the reference's recorder writes and its visualisers read (reference schema:
train/episode_record.py:117-156 `_schema`, row semantics :431-470 `_row`, file naming :530-545).

The reference builds one Python dict per row inside RLlib's callback.  Here a step of N markets arrives as
tensors (obs f32[N,168], reward f64[N,A], the SoA info tensors of include/cda.h `cda_info_ptrs`, the five action
tensors); the recorder keeps the selected markets' slices per step (one small device->host copy per tensor)
and assembles the Arrow columns array-wise when an episode ends - no per-row Python except the exact NAV string.

Column meaning (same names and types as the reference):
  identity   run_id, iteration, episode_id, step, agent_id ("agent_<i>"), module_id, wall_time, episode_complete
  NAV        nav (float of the exact value) and nav_str (`str(Decimal)`, what the conservation check parses)
  info       reward ... spread (spread / best_bid / best_ask are null on a one-sided book)
  reward_term_<nav_term|order_penalty|trade_penalty|drawdown_penalty|passive_bonus>
  action     [category, size_mean, size_sigma, price, price_offset] as float64 (the flattened action tuple)
  obs        the 168-float observation the agent saw after the step;  info_extra  null (every info key has a column)
"""
import os
import time

import numpy as np

from . import _capi as K

#: (column, arrow type name, source info tensor) in the reference's column order
INFO_COLUMNS = (
    ("reward", "float64", None), ("num_trades", "int64", "num_trades"), ("net_position", "int64", "net_position"),
    ("VWAP", "float64", "vwap"), ("cash", "float64", "cash"), ("cash_on_hold", "float64", "cash_on_hold"),
    ("position_val", "float64", "position_val"), ("drawdown", "float64", "drawdown"), ("max_nav", "float64", "max_nav"),
    ("num_trades_step", "int64", "num_trades_step"), ("num_passive_fills_step", "int64", "num_passive_fills_step"),
    ("order_step_placed", "int64", "order_step_placed"), ("num_rejected_step", "int64", "num_rejected_step"),
    ("is_pass_action", "bool", "is_pass_action"), ("last_price", "float64", "last_price"), ("best_bid", "float64", "best_bid"),
    ("best_ask", "float64", "best_ask"), ("spread", "float64", "spread"),
)
REWARD_TERMS = ("nav_term", "order_penalty", "trade_penalty", "drawdown_penalty", "passive_bonus")
_MARKET_LEVEL = ("last_price", "best_bid", "best_ask", "spread")


def schema():
    import pyarrow as pa
    kinds = {"float64": pa.float64(), "int64": pa.int64(), "bool": pa.bool_()}
    fields = [pa.field("run_id", pa.string()), pa.field("iteration", pa.int32()), pa.field("episode_id", pa.string()),
              pa.field("step", pa.int32()), pa.field("agent_id", pa.string()), pa.field("module_id", pa.string()),
              pa.field("wall_time", pa.float64()), pa.field("episode_complete", pa.bool_()),
              pa.field("nav", pa.float64()), pa.field("nav_str", pa.string())]
    fields += [pa.field(name, kinds[kind]) for name, kind, _ in INFO_COLUMNS]
    fields += [pa.field(f"reward_term_{t}", pa.float64()) for t in REWARD_TERMS]
    fields += [pa.field("action", pa.list_(pa.float64())), pa.field("obs", pa.list_(pa.float32())), pa.field("info_extra", pa.string())]
    return pa.schema(fields)


def _host(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class BatchedEpisodeRecorder:
    """Records the episodes of a chosen subset of markets of a batched env.

        rec = BatchedEpisodeRecorder(out_dir, num_agents=4, markets=[0, 17], run_id="r1")
        rec.begin_episodes(["ep-0", "ep-17"], module_ids=[["policy_0", ...], [...]])
        for t in range(T):
            obs, rew, term, trunc, info = env.step(*actions)       # env built with_info=True
            rec.record_step(obs, rew, info, actions)
        rec.finish()                                                # -> rows; a file once rows_per_file are pending
        rec.close()
    """

    def __init__(self, output_dir, num_agents, markets=(0,), run_id="", iteration=None, rows_per_file=65536, tag=None, sample_every=1):
        """sample_every (record_rollout): keep one episode in N, chosen by `zlib.crc32(str(episode id)) % N == 0` - the reference recorder's rule
        (train/episode_record.py:284-291; config/train_config.json episode_sample_every 10); 1 keeps everything."""
        self.sample_every = max(1, int(sample_every))
        self.output_dir = output_dir
        self.num_agents = int(num_agents)
        self.markets = np.asarray(list(markets), dtype=np.int64)
        self.run_id = run_id or ""
        self.iteration = iteration
        self.rows_per_file = max(1, int(rows_per_file))
        self.tag = tag or f"pid{os.getpid()}"
        self._seq = 0
        self._steps = []
        self._episode_ids = [f"market{int(m)}" for m in self.markets]
        self._module_ids = None
        self._pending = []
        self._pending_rows = 0
        self.written_rows = 0
        self.files = []

    # ------------------------------------------------------------------ sampling side
    def begin_episodes(self, episode_ids, module_ids=None):
        if len(episode_ids) != len(self.markets):
            raise ValueError("one episode id per recorded market")
        self._episode_ids = [str(e) for e in episode_ids]
        self._module_ids = None if module_ids is None else [[None if m is None else str(m) for m in row] for row in module_ids]
        self._steps = []

    def _take(self, x):
        """rows of the recorded markets from a full-batch tensor / array"""
        if hasattr(x, "index_select"):
            import torch
            idx = torch.as_tensor(self.markets, device=x.device)
            return x.index_select(0, idx).detach().cpu().numpy()
        return np.asarray(x)[self.markets]

    def record_step(self, obs, reward, info, actions, step_index=None):
        missing = [src for _, _, src in INFO_COLUMNS if src and src not in info] + [k for k in ("nav", "reward_terms") if k not in info]
        if missing:
            raise ValueError(f"the info tensors {missing} are needed (build the env with_info=True)")
        step = {"t": len(self._steps) if step_index is None else int(step_index), "wall": time.time(),
                "obs": self._take(obs).astype(np.float32, copy=False), "reward": self._take(reward).astype(np.float64, copy=False),
                "actions": [self._take(a) for a in actions]}
        for name in {src for _, _, src in INFO_COLUMNS if src} | {"nav", "reward_terms"}:
            step[name] = self._take(info[name])
        self._steps.append(step)

    def record_rollout(self, roll, iteration=None):
        """Feed the recorder from a fused rollout's buffers after the horizon (mlp.RolloutChains(info_markets=S): the last S markets of the env ran as a chain of
        their own with the info tensors of every step) - one device->host copy per tensor per ROLLOUT, no per-step host call.  `markets` must be those S markets.
        Episodes may span rollouts: rows accumulate until a step carries an episode end, which closes the episode (complete=True) and opens the next one
        (ids from `episode_namer(market, ordinal)`, module ids from `module_namer(market)` when set).  The last step of an episode is recorded with the episode's own
        last observation (the rollout's episode-end capture, when on) - obs[t + 1] already holds the next episode's first.  At every episode end the reference's
        conservation check runs on the recorded NAVs (league_based_self_play_callback.py:679-704: sum of NAV == agents x init_cash, in Decimal):
        `nav_checked` / `nav_violations` count episodes."""
        import decimal
        T, N, S, A = roll.T, roll.N, roll.info_markets, self.num_agents
        if roll.info is None or S != len(self.markets) or not np.array_equal(self.markets, np.arange(N - S, N)):
            raise ValueError("the recorder's markets must be the rollout's sampled chain (the last info_markets markets of the env)")
        if iteration is not None:
            self.iteration = iteration
        b = roll.buf
        host = lambda x: x[:, N - S:].detach().cpu().numpy()             # noqa: E731  [T, S, ...]
        obs_next = b["obs"][1:, N - S:].detach().cpu().numpy()             # [T, S, 168]: the observation after step t
        rew, term, trunc = host(b["reward"]), host(b["terminated"]), host(b["truncated"])
        acts = [host(b[k]) for k in ("category", "size_mean", "size_sigma", "price", "price_offset")]
        info = {k: v.detach().cpu().numpy() for k, v in roll.info.items()}   # already [T, S, ...]
        fin_idx = fin_obs = None
        if roll.capture_ends:
            fin_idx, fin_obs = host(b["fin_index"]), b["fin_obs"].detach().cpu().numpy()
        if not hasattr(self, "_ordinal"):
            self._init_market_state()
        wall = time.time()
        keys = sorted({src for _, _, src in INFO_COLUMNS if src} | {"nav", "reward_terms"})
        for t in range(T):
            o = obs_next[t].copy()
            done = (term[t] | trunc[t]).astype(bool)
            if fin_idx is not None:
                for j in np.nonzero(fin_idx[t] >= 0)[0]:
                    o[j] = fin_obs[fin_idx[t][j]]
            # every recorded market keeps its OWN episode: step counter, ordinal, rows - an env may end a market's episode early (all agents done,
            # done_helper.py:36-52) and the device-side auto reset then desynchronises the sampled markets; only the market whose `done` fired is closed, checked and renamed
            for j in range(S):
                step = {"t": int(self._mk_t[j]), "wall": wall, "obs": o[j:j + 1].astype(np.float32, copy=False), "reward": rew[t][j:j + 1].astype(np.float64, copy=False),
                        "actions": [a[t][j:j + 1] for a in acts]}
                for name in keys:
                    step[name] = info[name][t][j:j + 1]
                self._mk_steps[j].append(step)
                self._mk_t[j] += 1
                if done[j]:
                    if getattr(self, "init_cash", None) is not None:
                        nav = np.ascontiguousarray(info["nav"][t][j]).view(K.DEC_DTYPE).reshape(A)
                        with decimal.localcontext() as ctx:
                            ctx.prec = 28
                            total = sum((K.dec_to_decimal(nav[a]) for a in range(A)), decimal.Decimal(0))
                        self.nav_checked += 1
                        if abs(total - decimal.Decimal(A) * decimal.Decimal(int(self.init_cash))) > decimal.Decimal(str(getattr(self, "nav_tolerance", 1e-6))):
                            self.nav_violations += 1
                    self._finish_market(j, complete=True)
                    self._mk_ordinal[j] += 1
                    self._mk_t[j] = 0
                    self._name_market(j)
        self._ordinal = int(self._mk_ordinal.min())

    def _init_market_state(self):
        S = len(self.markets)
        self._ordinal = 0
        self._mk_ordinal, self._mk_t = np.zeros(S, np.int64), np.zeros(S, np.int64)
        self._mk_steps = [[] for _ in range(S)]
        self._mk_episode, self._mk_modules = [None] * S, [None] * S
        self.nav_checked = self.nav_violations = 0
        self._name_episodes()

    def _name_market(self, j):
        namer = getattr(self, "episode_namer", None) or (lambda m, k: f"market{m}-episode{k}")
        mods = getattr(self, "module_namer", None)
        m = int(self.markets[j])
        self._mk_episode[j] = str(namer(m, int(self._mk_ordinal[j])))
        self._mk_modules[j] = None if mods is None else [None if x is None else str(x) for x in mods(m)]

    def _name_episodes(self):
        """(re)name the episodes that have not recorded a step yet - the league calls this after it has drawn new opponents at an episode boundary"""
        if not hasattr(self, "_mk_t"):
            self._init_market_state()
            return
        for j in range(len(self.markets)):
            if self._mk_t[j] == 0:
                self._name_market(j)

    def _finish_market(self, j, complete):
        """close recorded market j's running episode: its rows become a table of their own (the column assembly of finish() on a one-market view)"""
        if not self._mk_steps[j]:
            return
        saved = (self._steps, self.markets, self._episode_ids, self._module_ids)
        try:
            self._steps, self.markets = self._mk_steps[j], self.markets[j:j + 1]
            self._episode_ids = [self._mk_episode[j]]
            self._module_ids = None if self._mk_modules[j] is None else [self._mk_modules[j]]
            self.finish(complete=complete)
        finally:
            self._steps, self.markets, self._episode_ids, self._module_ids = saved
            self._mk_steps[j] = []

    def sampled(self, episode_id):
        """the reference's sampling decision (train/episode_record.py:284-291): a pure function of the episode id"""
        import zlib
        return self.sample_every == 1 or zlib.crc32(str(episode_id).encode("utf-8")) % self.sample_every == 0

    def finish(self, complete=True):
        """The recorded markets' episodes ended (complete=False: sampling stopped before they did)."""
        if self._steps:
            table = self._table(complete)
            if self.sample_every > 1:                      # keep the rows of the sampled episodes only
                import pyarrow as pa
                import pyarrow.compute as pc
                keep = [e for e in self._episode_ids if self.sampled(e)]
                table = table.filter(pc.is_in(table.column("episode_id"), value_set=pa.array(keep, pa.string()))) if keep else table.slice(0, 0)
            if table.num_rows:
                self._pending.append(table)
                self._pending_rows += table.num_rows
            self._steps = []
        if self._pending_rows >= self.rows_per_file:
            self._write()

    def flush(self):
        """close what is still running as incomplete (sampling stopped before the episodes did) and write the pending rows"""
        for j in range(len(getattr(self, "_mk_steps", []))):
            self._finish_market(j, complete=False)
        if self._steps:
            self.finish(complete=False)
        return self._write()

    def _write(self):
        import pyarrow as pa
        import pyarrow.parquet as pq
        if not self._pending:
            return None
        os.makedirs(self.output_dir, exist_ok=True)
        self._seq += 1
        path = os.path.join(self.output_dir, f"episodes.{self.tag}.{self._seq:06d}.parquet")
        table = pa.concat_tables(self._pending)
        pq.write_table(table, path, compression="snappy")
        self.written_rows += table.num_rows
        self.files.append(path)
        self._pending, self._pending_rows = [], 0
        return path

    def close(self):
        return self.flush()

    # ------------------------------------------------------------------ column assembly
    def _table(self, complete):
        import pyarrow as pa
        T, M, A = len(self._steps), len(self.markets), self.num_agents
        R = M * T * A                                         # row order: market (episode), step, agent

        def per_agent(key, dtype=None):                        # [T][M, A, ...] -> [M, T, A, ...] -> rows
            x = np.stack([s[key] for s in self._steps], axis=1)
            x = x.reshape((R,) + x.shape[3:])
            return x if dtype is None else x.astype(dtype)

        def per_market(key):                                   # [T][M] -> broadcast over agents
            x = np.stack([s[key] for s in self._steps], axis=1)            # [M, T]
            return np.repeat(x.reshape(M * T), A)

        def nullable(x):
            return pa.array(x, mask=np.isnan(x))

        cols = {}
        cols["run_id"] = pa.array([self.run_id] * R, pa.string())
        cols["iteration"] = pa.array([self.iteration] * R, pa.int32())
        cols["episode_id"] = pa.array(np.repeat(np.array(self._episode_ids, dtype=object), T * A), pa.string())
        cols["step"] = pa.array(np.tile(np.repeat(np.array([s["t"] for s in self._steps], np.int32), A), M))
        cols["agent_id"] = pa.array(np.tile(np.array([f"agent_{a}" for a in range(A)], dtype=object), M * T), pa.string())
        if self._module_ids is None:
            cols["module_id"] = pa.array([None] * R, pa.string())
        else:
            mods = np.array(self._module_ids, dtype=object)                 # [M, A]
            cols["module_id"] = pa.array(np.broadcast_to(mods[:, None, :], (M, T, A)).reshape(R), pa.string())
        cols["wall_time"] = pa.array(np.tile(np.repeat(np.array([s["wall"] for s in self._steps], np.float64), A), M))
        cols["episode_complete"] = pa.array(np.full(R, bool(complete)))
        nav_raw = per_agent("nav")
        nav_rec = np.ascontiguousarray(nav_raw).view(K.DEC_DTYPE).reshape(R)
        nav_dec = [K.dec_to_decimal(r) for r in nav_rec]
        cols["nav"] = pa.array(np.array([float(d) for d in nav_dec], np.float64))
        cols["nav_str"] = pa.array([str(d) for d in nav_dec], pa.string())
        for name, kind, src in INFO_COLUMNS:
            if name == "reward":
                cols[name] = pa.array(per_agent("reward", np.float64))
            elif name in _MARKET_LEVEL:
                x = per_market(src).astype(np.float64)
                cols[name] = nullable(x)                       # NaN on the device = None in the reference's info dict
            elif kind == "int64":
                cols[name] = pa.array(per_agent(src, np.int64))
            elif kind == "bool":
                cols[name] = pa.array(per_agent(src).astype(bool))
            else:
                cols[name] = pa.array(per_agent(src, np.float64))
        terms = per_agent("reward_terms", np.float64)          # [R, 5]
        for i, t in enumerate(REWARD_TERMS):
            cols[f"reward_term_{t}"] = pa.array(np.ascontiguousarray(terms[:, i]))
        acts = np.stack([np.stack([a.astype(np.float64) for a in s["actions"]], axis=-1) for s in self._steps], axis=1)   # [M, T, A, 5]
        cols["action"] = pa.ListArray.from_arrays(pa.array(np.arange(0, 5 * R + 1, 5, dtype=np.int32)), pa.array(acts.reshape(-1)))
        obs = np.stack([s["obs"] for s in self._steps], axis=1)                 # [M, T, D]
        D = obs.shape[-1]
        obs_rows = np.repeat(obs.reshape(M * T, D), A, axis=0).astype(np.float32)   # every agent of a market sees the same vector
        cols["obs"] = pa.ListArray.from_arrays(pa.array(np.arange(0, D * R + 1, D, dtype=np.int32)), pa.array(obs_rows.reshape(-1)))
        cols["info_extra"] = pa.array([None] * R, pa.string())
        sch = schema()
        return pa.Table.from_arrays([cols[f.name] for f in sch], schema=sch)
