#!/usr/bin/env python3
"""A SHORT one-chain policy rollout (bench.py's value_policy_in_loop at the driver's 20 steps): one HIP graph replay against the same launches issued directly by
cda_mlp_rollout_chain (one C call, T launches).  K = 20 / 64 steps, 4096 x 4 and 2048 x 8.   python tools/policy_leg_graph_vs_direct.py"""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_continuousdoubleauction_amd import CDAVecEnv, mlp  # noqa: E402


def timed(fn, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3, min(ts) * 1e3


for N, A in ((4096, 4), (2048, 8)):
    for K in (20, 64):
        row = {"markets": N, "agents": A, "steps": K}
        for graphs in (True, False, True, False):
            env = CDAVecEnv({"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False, "auto_reset": True}, n_markets=N, with_info=False)
            env.reset(seed=1000)
            pol = mlp.FusedPolicy(torch.device("cuda:0"), seed=0)
            roll = mlp.RolloutChains(env, pol, K, groups=1, seed=2024, use_graphs=graphs)
            for _ in range(3):
                roll.run()
            med, best = timed(roll.run)
            row.setdefault("graph" if graphs else "direct", []).append(round(N * A * K / med / 1e3, 1))
            del roll
            env.close()
        print(json.dumps(row), flush=True)
