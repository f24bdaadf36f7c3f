"""Shared by the CPU and GPU episode-record tests: replay the first steps of the trace_A4_s0 golden through a
stepper (numpy adapter over the oracle or over the HIP env), record them, and compare the Parquet file with the
one the REFERENCE's recorder wrote for the same episode (tests/golden/episode_record_ref.parquet)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
T = 24
SKIP = {"run_id", "wall_time", "iteration"}           # identity / clock columns, not part of the episode's content


def replay_and_record(make_env, out_dir):
    """make_env(config, n_markets) -> object with reset(seeds) and step(cat, mean, sigma, price, off) ->
    (obs, reward, term, trunc, info) for n_markets markets (market 1 carries the golden episode, 0 and 2 are decoys)."""
    from gym_continuousdoubleauction_amd.episode_record import BatchedEpisodeRecorder
    tr = np.load(os.path.join(HERE, "golden", "trace_A4_s0.npz"), allow_pickle=True)
    cfg = json.loads(str(tr["config"]))
    A = int(cfg["num_of_agents"])
    env = make_env(cfg, 3)
    seed = int(tr["seed"])
    env.reset(np.array([seed + 1, seed, seed + 2], dtype=np.uint64))
    rec = BatchedEpisodeRecorder(out_dir, num_agents=A, markets=[1], run_id="test")
    rec.begin_episodes(["trace_A4_s0"], module_ids=[[f"policy_{a}" for a in range(A)]])
    rng = np.random.default_rng(3)
    for t in range(T):
        def batch(x, lo, hi, dt):
            full = rng.integers(lo, hi, (3, A)).astype(dt) if np.issubdtype(dt, np.integer) else rng.uniform(lo, hi, (3, A)).astype(dt)
            full[1] = x[t]
            return full
        acts = (batch(tr["cat"], 0, 9, np.int32), batch(tr["mean"], -1, 1, np.float32), batch(tr["sigma"], 0, 1, np.float32),
                batch(tr["price"], 0, 10, np.int32), batch(tr["off"], 0, 3, np.int32))
        obs, rew, term, trunc, info = env.step(*acts)
        rec.record_step(obs, rew, info, acts)
    rec.finish(complete=True)
    path = rec.close()
    assert rec.written_rows == T * A
    return path


def assert_same_as_reference(path):
    import pyarrow.parquet as pq
    got, ref = pq.read_table(path), pq.read_table(os.path.join(HERE, "golden", "episode_record_ref.parquet"))
    assert got.schema.equals(ref.schema), (got.schema, ref.schema)
    assert got.num_rows == ref.num_rows
    for name in ref.schema.names:
        if name in SKIP:
            continue
        g, r = got.column(name).to_pylist(), ref.column(name).to_pylist()
        if name in ("obs", "action") or ref.schema.field(name).type in ("double",):
            pass
        assert len(g) == len(r)
        for i, (x, y) in enumerate(zip(g, r)):
            if isinstance(y, float):
                assert x is not None and np.float64(x).tobytes() == np.float64(y).tobytes(), (name, i, x, y)
            elif isinstance(y, list):
                assert np.array_equal(np.asarray(x, np.float64).view(np.uint64), np.asarray(y, np.float64).view(np.uint64)), (name, i)
            else:
                assert x == y, (name, i, x, y)
