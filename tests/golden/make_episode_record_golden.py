#!/usr/bin/env python3
"""Cut the golden Parquet file of the per-step episode record from the REFERENCE: its env
(envs/continuousDoubleAuction_env.py) stepped on the action stream of the committed trace_A4_s0 golden, its own
recorder (train/episode_record.py EpisodeRecorder) fed through a minimal episode object.  Build container only.

    python tests/golden/make_episode_record_golden.py   -> tests/golden/episode_record_ref.parquet
"""
import glob
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(0, "/root/reference")
from gym_continuousDoubleAuction.envs.continuousDoubleAuction_env import continuousDoubleAuctionEnv  # noqa: E402
from gym_continuousDoubleAuction.train.episode_record import EpisodeRecorder  # noqa: E402

T = 24


class _Episode:
    """What the recorder reads from an RLlib episode: id_, the last step's dicts, module_for."""

    def __init__(self, episode_id):
        self.id_ = episode_id
        self.last = None

    def get_infos(self, index):
        return self.last[3]

    def get_observations(self, index):
        return self.last[0]

    def get_actions(self, index):
        return self.last[1]

    def get_rewards(self, index):
        return self.last[2]

    def module_for(self, agent_id):
        return "policy_" + agent_id.split("_")[1]


def main():
    tr = np.load(os.path.join(HERE, "trace_A4_s0.npz"), allow_pickle=True)
    cfg = json.loads(str(tr["config"]))
    env = continuousDoubleAuctionEnv(dict(cfg))
    A = env.num_of_agents
    env.reset(seed=int(tr["seed"]))
    out_dir = tempfile.mkdtemp()
    rec = EpisodeRecorder(out_dir, run_id="golden")
    ep = _Episode("trace_A4_s0")
    for t in range(T):
        actions, tuples = {}, {}
        for a in range(A):
            actions[f"agent_{a}"] = {"category": np.int64(tr["cat"][t, a]), "size_mean": np.array([tr["mean"][t, a]], np.float32),
                                     "size_sigma": np.array([tr["sigma"][t, a]], np.float32), "price": np.int64(tr["price"][t, a]),
                                     "price_offset": np.int64(tr["off"][t, a])}
            tuples[f"agent_{a}"] = (np.int32(tr["cat"][t, a]), np.array([tr["mean"][t, a]], np.float32), np.array([tr["sigma"][t, a]], np.float32),
                                    np.int32(tr["price"][t, a]), np.int32(tr["off"][t, a]))
        obs, rewards, terms, truncs, infos = env.step(actions)
        assert np.array_equal(obs["agent_0"], tr["obs"][t])          # the trace and this run are the same episode
        ep.last = (obs, tuples, rewards, infos)
        rec.record_step(ep, t)
    rec.finish_episode(ep.id_)
    rec.close()
    files = glob.glob(os.path.join(out_dir, "*.parquet"))
    assert len(files) == 1, files
    dst = os.path.join(HERE, "episode_record_ref.parquet")
    shutil.copyfile(files[0], dst)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
