"""What does a step cost as a function of what the agents do?  (run on the GPU box)  One launch per step (groups=1),
4096 x 4, books warmed by 300 random steps; then 100 steps of ONE action category for every agent, timed with events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402

N, A = 4096, 4
dev = torch.device("cuda:0")
cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}
NAMES = {0: "pass", 1: "bid market", 2: "bid limit", 3: "bid modify", 4: "bid cancel", 5: "ask market", 6: "ask limit", 7: "ask modify", 8: "ask cancel", 9: "RANDOM (uniform law)"}
for info in (False,):
    for cat_v in (9, 0, 2, 6, 3, 4, 1, 5):
        env = CDAVecEnv(cfg, N, with_info=info)
        env.reset(seed=1000)
        acts = env.random_actions_device(0, 400, action_seed=2024)
        for t in range(300):
            env.step(*[x[t] for x in acts])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 100
        torch.cuda.synchronize()
        a.record()
        for t in range(300, 300 + steps):
            cat = acts[0][t] if cat_v == 9 else torch.full_like(acts[0][t], cat_v)
            env.step(cat, acts[1][t], acts[2][t], acts[3][t], acts[4][t])
        b.record()
        torch.cuda.synchronize()
        print(f"{NAMES[cat_v]:24s} {a.elapsed_time(b) / steps * 1000:7.1f} us/step   (peak book {int(env.book_peak().max())})", flush=True)
        env.close()
