"""The per-step episode record (gym_continuousdoubleauction_amd/episode_record.py) against the Parquet file the
reference's own recorder wrote for the same episode.  CPU: the oracle is the stepper; GPU: the HIP env."""
import os

import numpy as np
import pytest

import record_util as R


def test_schema_is_the_reference_schema():
    import pyarrow.parquet as pq
    from gym_continuousdoubleauction_amd.episode_record import schema
    ref = pq.read_schema(os.path.join(R.HERE, "golden", "episode_record_ref.parquet"))
    assert schema().equals(ref)


def test_record_from_oracle_stepper_equals_reference_file(tmp_path):
    import oracle_lib as O
    path = R.replay_and_record(lambda cfg, n: O.OracleEnv(cfg, n_markets=n), str(tmp_path))
    R.assert_same_as_reference(path)


def test_incomplete_flag_and_file_rotation(tmp_path):
    import pyarrow.parquet as pq
    from gym_continuousdoubleauction_amd.episode_record import BatchedEpisodeRecorder
    import oracle_lib as O
    cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 64, "is_render": False}
    env = O.OracleEnv(cfg, n_markets=5)
    env.reset(np.arange(5, dtype=np.uint64))
    rec = BatchedEpisodeRecorder(str(tmp_path), num_agents=4, markets=[0, 4], rows_per_file=40)
    rng = np.random.default_rng(0)
    for ep in range(3):
        rec.begin_episodes([f"e{ep}-m0", f"e{ep}-m4"])
        for t in range(4):
            acts = (rng.integers(0, 9, (5, 4)).astype(np.int32), rng.uniform(-1, 1, (5, 4)).astype(np.float32), rng.uniform(0, 1, (5, 4)).astype(np.float32),
                    rng.integers(0, 10, (5, 4)).astype(np.int32), rng.integers(0, 3, (5, 4)).astype(np.int32))
            obs, rew, term, trunc, info = env.step(*acts)
            rec.record_step(obs, rew, info, acts)
        if ep < 2:
            rec.finish(complete=True)
    rec.close()                                             # the third pair of episodes never ended
    assert len(rec.files) == 2 and rec.written_rows == 3 * 2 * 4 * 4
    t = pq.read_table(rec.files).to_pandas()
    assert set(t["episode_id"]) == {f"e{e}-m{m}" for e in range(3) for m in (0, 4)}
    assert t[t.episode_id.str.startswith("e2")]["episode_complete"].eq(False).all()
    assert t[~t.episode_id.str.startswith("e2")]["episode_complete"].all()
    assert t["module_id"].isna().all() and t["info_extra"].isna().all()
    one = t[(t.episode_id == "e1-m4") & (t.agent_id == "agent_2")]
    assert list(one["step"]) == [0, 1, 2, 3] and all(len(o) == 168 for o in one["obs"]) and all(len(a) == 5 for a in one["action"])


@pytest.mark.gpu
def test_record_from_hip_env_equals_reference_file(tmp_path):
    from gym_continuousdoubleauction_amd import CDAVecEnv

    class TensorStepper:                                     # device tensors go straight into the recorder
        def __init__(self, cfg, n):
            self.env = CDAVecEnv(cfg, n_markets=n, with_info=True)

        def reset(self, seeds):
            return self.env.reset(seed=seeds)

        def step(self, *acts):
            return self.env.step(*acts)

    path = R.replay_and_record(TensorStepper, str(tmp_path))
    R.assert_same_as_reference(path)


def test_sample_every_keeps_the_episodes_the_reference_rule_names(tmp_path):
    """One episode in N by crc32 of the episode id (train/episode_record.py:284-291): the same subset on every worker, no coordination."""
    import zlib
    import pyarrow.parquet as pq
    from gym_continuousdoubleauction_amd.episode_record import BatchedEpisodeRecorder
    import oracle_lib as O
    cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 64, "is_render": False}
    env = O.OracleEnv(cfg, n_markets=6)
    env.reset(np.arange(6, dtype=np.uint64))
    rec = BatchedEpisodeRecorder(str(tmp_path), num_agents=4, markets=range(6), sample_every=3)
    rng = np.random.default_rng(0)
    ids = []
    for ep in range(4):
        names = [f"run-episode{ep}-market{m}" for m in range(6)]
        ids += names
        rec.begin_episodes(names)
        for t in range(3):
            acts = (rng.integers(0, 9, (6, 4)).astype(np.int32), rng.uniform(-1, 1, (6, 4)).astype(np.float32), rng.uniform(0, 1, (6, 4)).astype(np.float32),
                    rng.integers(0, 10, (6, 4)).astype(np.int32), rng.integers(0, 3, (6, 4)).astype(np.int32))
            obs, rew, term, trunc, info = env.step(*acts)
            rec.record_step(obs, rew, info, acts)
        rec.finish(complete=True)
    path = rec.close()
    want = {e for e in ids if zlib.crc32(e.encode()) % 3 == 0}
    assert 0 < len(want) < len(ids)
    t = pq.read_table(path).to_pandas()
    assert set(t["episode_id"]) == want and len(t) == len(want) * 3 * 4


def test_record_rollout_keeps_one_episode_per_market_when_the_markets_desynchronise(tmp_path):
    """Round-5 ADVICE: an env may end a market's episode early (all agents done, done_helper.py:36-52); the device-side auto reset then desynchronises the sampled
    markets.  record_rollout closes, checks and renames ONLY the market whose `done` fired; the others' episodes run on with their own step counters.  (A stand-in for
    mlp.RolloutChains with host tensors: record_rollout reads buffers only.)"""
    import decimal
    import types

    import pyarrow.parquet as pq
    import torch
    from gym_continuousdoubleauction_amd import _capi as K
    from gym_continuousdoubleauction_amd.episode_record import BatchedEpisodeRecorder
    N, S, A, T = 5, 2, 3, 6
    rng = np.random.default_rng(4)
    ends = {(2, 0), (5, 0), (4, 1)}                                  # (step, recorded market): market 0 ends twice, market 1 once - never together
    term = np.zeros((T, N), np.uint8); trunc = np.zeros((T, N), np.uint8)
    for t, j in ends:
        (term if (t, j) == (2, 0) else trunc)[t, N - S + j] = 1
    nav = np.zeros((T, S, A), K.DEC_DTYPE)
    for t in range(T):
        for j in range(S):
            vals = [decimal.Decimal(1000), decimal.Decimal("999.5"), decimal.Decimal("1000.5")]
            if (t, j) == (4, 1):
                vals[2] += decimal.Decimal("0.25")                    # a ledger fault at market 1's episode end
            for a in range(A):
                nav[t, j, a] = np.asarray(K.decimal_to_dec(vals[a])).view(K.DEC_DTYPE).reshape(())
    info = {}
    for name, ct, per_agent, dims in K.INFO_FIELDS:
        shape = (T, S) + ((A,) if per_agent else ()) + tuple(dims)
        if name == "nav":
            info[name] = torch.from_numpy(nav.view(np.uint8).reshape(T, S, A, 16).copy())
        elif ct is K.C.c_double:
            info[name] = torch.from_numpy(rng.normal(size=shape))
        else:
            info[name] = torch.from_numpy(rng.integers(0, 3, shape).astype(np.uint8 if ct is K.C.c_uint8 else np.int32))
    buf = {"obs": torch.from_numpy(rng.normal(size=(T + 1, N, 168)).astype(np.float32)), "reward": torch.from_numpy(rng.normal(size=(T, N, A))),
           "terminated": torch.from_numpy(term), "truncated": torch.from_numpy(trunc)}
    for k, dt in (("category", np.int32), ("size_mean", np.float32), ("size_sigma", np.float32), ("price", np.int32), ("price_offset", np.int32)):
        buf[k] = torch.from_numpy(rng.integers(0, 3, (T, N, A)).astype(dt))
    roll = types.SimpleNamespace(T=T, N=N, info_markets=S, info=info, buf=buf, capture_ends=False)
    rec = BatchedEpisodeRecorder(str(tmp_path), num_agents=A, markets=range(N - S, N), run_id="d", rows_per_file=10 ** 9)
    rec.init_cash = 1000
    rec.record_rollout(roll, iteration=0)
    t = pq.read_table(rec.close()).to_pandas()
    per = {e: g for e, g in t[t.agent_id == "agent_0"].groupby("episode_id")}
    assert {e: (list(g["step"]), bool(g["episode_complete"].iloc[0])) for e, g in per.items()} == {
        "market3-episode0": ([0, 1, 2], True), "market3-episode1": ([0, 1, 2], True),               # steps 0-2 and 3-5 of the rollout
        "market4-episode0": ([0, 1, 2, 3, 4], True), "market4-episode1": ([0], False)}                # ... 0-4; the episode begun at step 5 never ended
    assert rec.nav_checked == 3 and rec.nav_violations == 1                                           # only the markets whose episode ended are checked
    assert t.shape[0] == T * S * A
