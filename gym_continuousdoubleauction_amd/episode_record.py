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

    def __init__(self, output_dir, num_agents, markets=(0,), run_id="", iteration=None, rows_per_file=65536, tag=None):
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

    def finish(self, complete=True):
        """The recorded markets' episodes ended (complete=False: sampling stopped before they did)."""
        if self._steps:
            self._pending.append(self._table(complete))
            self._pending_rows += self._pending[-1].num_rows
            self._steps = []
        if self._pending_rows >= self.rows_per_file:
            self.flush()

    def flush(self):
        import pyarrow as pa
        import pyarrow.parquet as pq
        if self._steps:
            self.finish(complete=False)
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
