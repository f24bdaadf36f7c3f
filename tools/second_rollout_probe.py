#!/usr/bin/env python3
"""Why is the SECOND rollout of ppo.train_fused slow (38-45 ms against 3.6 ms) since the history-depth variants?  Host timers around the pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
env = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}, n_markets=4096, with_info=False)
dev = env.obs.device
pol = mlp.FusedPolicy(dev, seed=0)
env.reset(seed=0)
roll = mlp.RolloutChains(env, pol, 64, groups=4, seed=0)
upd = mlp.FusedUpdate(pol, 64 * 4096, 65536, 4)
ret = mlp.EpisodeReturns(4096, 4, dev)
def t(fn, name):
    torch.cuda.synchronize(); a = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print(f"  {name}: {(time.perf_counter() - a) * 1e3:.2f} ms"); return r
for it in range(4):
    print("iteration", it)
    buf = t(roll.run, "run")
    rec = t(lambda: roll.gae(), "gae")
    if os.environ.get("SKIP_UPDATE") != "1":
        t(lambda: upd.run(buf["obs"][:64].view(64 * 4096, -1), records=rec), "update")
    if os.environ.get("SKIP_RET") != "1":
        t(lambda: ret.update(buf, 64).cpu(), "episode returns")
        t(lambda: float(buf["reward"].mean()), "reward mean")
