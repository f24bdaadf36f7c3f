"""Where the episode metrics cost: the per-step tallies, the episode-end path, the collection.  python tools/episode_metrics_probe.py [markets agents]"""
import json
import statistics
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from gym_continuousdoubleauction_amd import CDAVecEnv, mlp  # noqa: E402


def timed(fn, n=7):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    return statistics.median(out) * 1e3


def main():
    N, A = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4)
    T = 256
    res = {"markets": N, "agents": A, "horizon": T}
    pol = mlp.FusedPolicy("cuda:0", seed=0)
    for max_step in (1 << 20, 64, 16):
        env = CDAVecEnv({"num_of_agents": A, "init_cash": 1000000, "max_step": max_step, "is_render": False, "auto_reset": True}, n_markets=N, with_info=False)
        env.reset(seed=1)
        row = {}
        for on in (False, True, False, True):
            env.enable_episode_metrics(on)
            roll = mlp.RolloutChains(env, pol, T, groups=4, seed=3)
            for _ in range(3):
                roll.run()
            row.setdefault("on" if on else "off", []).append(timed(roll.run))
            del roll
        env.enable_episode_metrics(True)
        row["collect_ms"] = timed(lambda: env.collect_episode_metrics())
        out = (torch.empty((1, 32), dtype=torch.float64, device="cuda:0"), torch.empty(8, dtype=torch.float64, device="cuda:0"))
        row["collect_reused_buffers_ms"] = timed(lambda: env.collect_episode_metrics(out=out))
        res[f"max_step_{max_step}"] = {"rollout_ms_off": row["off"], "rollout_ms_on": row["on"], "on_over_off": statistics.mean(row["on"]) / statistics.mean(row["off"]),
                                       "collect_ms": row["collect_ms"], "collect_reused_buffers_ms": row["collect_reused_buffers_ms"]}
        env.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
