"""What the general build (book beyond the LDS tile, HBM tier in play) costs per step (GPU box):
4096 markets under the drifting "trend" law of the big-book goldens, timed in windows as the books grow.
    python tools/bigbook_speed.py > profiles/r03/bigbook_speed.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from book_census import trend  # noqa: E402
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402

dev = torch.device("cuda:0")
rows = []
for a in (4, 16):
    n, T, W = 4096, 1024, 128
    env = CDAVecEnv({"num_of_agents": a, "init_cash": 1000000, "max_step": T + 1, "is_render": False}, n, with_info=False)
    env.reset(seed=1000)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    acts = [trend(gen, n, a, dev) for _ in range(W)]                  # one window of resident actions, replayed
    for w in range(T // W):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for x in acts:
            env.step(*x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / W
        peak = env.book_peak()
        rows.append({"agents": a, "steps_done": (w + 1) * W, "us_per_step": dt * 1e6, "agent_steps_per_s": n * a / dt,
                     "median_resting_orders": float(peak.float().median()), "max_resting_orders": int(peak.max()),
                     "markets_beyond_tile": int((peak > env.book_capacity).sum()), "tile": env.book_capacity})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    assert int((env.flags() != 0).sum()) == 0 and int((env.check_invariants() != 0).sum()) == 0
    env.close()
print(json.dumps({"workload": "4096 markets, trend law (tests/golden/make_goldens.py), one launch per step, windows of 128 steps", "rows": rows}, indent=1))
