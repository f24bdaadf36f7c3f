#!/usr/bin/env python3
"""Where does a SHORT policy-in-the-loop rollout (bench.py's value_policy_in_loop at the driver's 20 steps) lose against a long one?
Times RolloutChains.run() sync to sync for K = 20 / 64 / 256 steps and 1 / 2 / 4 chains, the host's enqueue time, every chain's start and finish
(round 6, first version of this probe, profiles/r06/policy_leg_probe_before.jsonl: as shipped then / without the counter increment ahead of the fork / bare graph
replays / the chains captured into ONE graph - the counter kernel cost 20-30 us, fork + join 30-50 us, every further chain starts ~35 us after its predecessor,
one multi-branch graph is slower than four).  Now: the shipped sequence (per-chain counters bumped in the chain, one chain = the caller's stream)."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_continuousdoubleauction_amd import CDAVecEnv, mlp  # noqa: E402


def timed(fn, reps=9):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0, th))
    return statistics.median(t for t, _ in ts) * 1e3, min(t for t, _ in ts) * 1e3, statistics.median(h for _, h in ts) * 1e3


def main():
    N, A = int(os.environ.get("PROBE_MARKETS", 4096)), int(os.environ.get("PROBE_AGENTS", 4))
    dev = torch.device("cuda:0")
    out = []
    for K in (20, 64, 256):
        for chains in (4, 2, 1):
            env = CDAVecEnv({"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False, "auto_reset": True}, n_markets=N, with_info=False)
            env.reset(seed=1000)
            pol = mlp.FusedPolicy(dev, seed=0)
            roll = mlp.RolloutChains(env, pol, K, groups=chains, seed=2024, use_graphs=True)
            for _ in range(3):
                roll.run()
            r = {"steps": K, "chains": chains, "markets": N, "agents": A}
            med, best, host = timed(roll.run)
            r["shipped"] = {"ms": med, "best_ms": best, "host_enqueue_ms": host, "us_per_step": med / K * 1e3, "M_agent_steps_per_s": N * A * K / med / 1e3}
            print(json.dumps(r), flush=True)
            out.append(r)
            del roll
            env.close()
    return out


if __name__ == "__main__":
    main()
