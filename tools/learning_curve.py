#!/usr/bin/env python3
"""Does the fused PPO loop learn?  Short episodes (max_step == horizon: every iteration sees whole episodes of the same phase), the fused loop
(ppo.train_fused, hand-written bf16 MFMA kernels) and the legacy float32 torch loop (ppo.train) on the same seeds and hyper-parameters; prints one JSON line
per loop with the mean episode return per iteration (per agent).  tests/test_hip_learning.py pins what this measured.

    python tools/learning_curve.py --markets 1024 --agents 4 --episode 32 --iters 40 --lr 3e-4 [--objective rllib]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def curves(markets=1024, agents=4, episode=32, iters=40, lr=3e-4, seed=0, objective=None, legacy=True, minibatch=None, log_std_head=False):
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    cfg = {"num_of_agents": agents, "init_cash": 1000000, "max_step": episode, "is_render": False, "auto_reset": True}
    minibatch = minibatch or markets * episode * agents // 2
    out = {}
    env = CDAVecEnv(cfg, n_markets=markets, with_info=False)
    _, hist = ppo.train_fused(env, iters=iters, horizon=episode, lr=lr, seed=seed, log=lambda s: None, minibatch=minibatch, objective=objective, state_dependent_log_std=log_std_head)
    out["fused"] = [h["episode_return"] for h in hist]
    out["fused_entropy"] = [h["entropy"] for h in hist]
    out["fused_kl"] = [h["kl"] for h in hist]
    out["fused_v_loss"] = [h["v_loss"] for h in hist]
    env.close()
    if legacy:
        env = CDAVecEnv(cfg, n_markets=markets, with_info=False)
        _, hist = ppo.train(env, iters=iters, horizon=episode, lr=lr, seed=seed, log=lambda s: None, amp=False)
        out["legacy"] = [h["mean_reward"] * episode for h in hist]          # whole episodes per iteration: mean reward x steps = the episode return per agent
        env.close()
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--markets", type=int, default=1024)
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--episode", type=int, default=32)
    p.add_argument("--iters", type=int, default=40)
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--objective", choices=("ppo", "rllib"), default="ppo")
    p.add_argument("--no-legacy", action="store_true")
    p.add_argument("--log-std-head", action="store_true", help="the fused loop's policy carries the state-dependent log-std head")
    a = p.parse_args()
    from gym_continuousdoubleauction_amd import ppo
    c = curves(a.markets, a.agents, a.episode, a.iters, a.lr, a.seed, ppo.RLLIB_DEFAULTS if a.objective == "rllib" else None, legacy=not a.no_legacy, log_std_head=a.log_std_head)
    r = lambda xs: [None if x is None else round(float(x), 4) for x in xs]          # noqa: E731
    print(json.dumps({"config": vars(a), **{k: r(v) for k, v in c.items()}}))
    f = c["fused"]
    print(json.dumps({"fused_first3": sum(f[:3]) / 3, "fused_last3": sum(f[-3:]) / 3,
                      **({"legacy_first3": sum(c["legacy"][:3]) / 3, "legacy_last3": sum(c["legacy"][-3:]) / 3} if "legacy" in c else {})}))


if __name__ == "__main__":
    main()
