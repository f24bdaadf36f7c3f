#!/usr/bin/env python3
"""Host-side cost of CDAVecEnv.step(pipelined=True) per step by number of group chains: the time to ENQUEUE a burst of 40 steps from an idle device (the GPU runs
behind; nothing waits for it) - what bounds short free-running legs once chains x host-cost-per-launch exceeds a launch's duration.   python tools/step_enqueue_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402

N, A = 4096, 4
cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 100000, "is_render": False}
for info in (True, False):
    for G in (1, 2, 4, 6):
        env = CDAVecEnv(cfg, n_markets=N, with_info=info, groups=G)
        env.reset(seed=1000)
        acts = env.random_actions_device(0, 64, action_seed=1)
        steps = [tuple(a[i] for a in acts) for i in range(64)]
        for i in range(64):
            env.step(*steps[i], pipelined=True)
        best = 1e9
        for _ in range(5):
            env.sync()
            t0 = time.perf_counter()
            for i in range(40):
                env.step(*steps[i], pipelined=True)
            best = min(best, (time.perf_counter() - t0) / 40)
            env.sync()
        print(f"info {int(info)}  {G} chain(s): {best * 1e6:6.1f} us of host time per step() = {best * 1e6 / G:5.1f} per launch")
        env.close()
