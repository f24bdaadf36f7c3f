#!/usr/bin/env python3
"""What does recording episodes from the fused rollout cost?  (rollout cost <= +3 %.)  The PPO loop of BASELINE configs[4] (4096 markets x 4 agents,
horizon 64) with and without a sampled chain of S markets that writes the info tensors of every step + the recorder fed from the rollout buffers after the
horizon (episode_record.BatchedEpisodeRecorder.record_rollout).  Device times of rollout and update per iteration (median of the iterations after the first),
and the host time the recorder spends per rollout OUTSIDE them (device->host copies of the sampled markets' slices + the Arrow columns at episode ends).

    python tools/record_cost_probe.py [--sampled 8] [--iters 10] > profiles/r05/record_cost.txt
"""
import argparse
import os
import statistics
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run(markets, agents, episode, horizon, iters, sampled, league):
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    from gym_continuousdoubleauction_amd.episode_record import BatchedEpisodeRecorder
    from gym_continuousdoubleauction_amd.league_train import train_league_fused
    cfg = {"num_of_agents": agents, "init_cash": 1000000, "max_step": episode, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=markets, with_info=False)
    rec, host = None, []
    tmp = tempfile.mkdtemp()
    if sampled:
        rec = BatchedEpisodeRecorder(tmp, num_agents=agents, markets=range(markets - sampled, markets), run_id="probe")
        rec.init_cash = 1000000
        inner = rec.record_rollout

        def timed_record(roll, iteration=None):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            inner(roll, iteration=iteration)
            host.append(time.perf_counter() - t0)
        rec.record_rollout = timed_record
    if league:
        _, _, hist = train_league_fused(env, iters=iters, horizon=horizon, num_trainable=2, log=lambda s: None, recorder=rec, info_markets=sampled)
    else:
        _, hist = ppo.train_fused(env, iters=iters, horizon=horizon, log=lambda s: None, recorder=rec, info_markets=sampled)
    if rec is not None:
        rec.close()
    env.close()
    tail = hist[2:]
    out = {"rollout_ms": statistics.median(h["rollout_s"] for h in tail) * 1e3, "update_ms": statistics.median(h["update_s"] for h in tail) * 1e3,
           "host_record_ms": statistics.median(host[2:]) * 1e3 if host else 0.0, "rows": rec.written_rows if rec else 0,
           "nav_checked": getattr(rec, "nav_checked", 0), "nav_violations": getattr(rec, "nav_violations", 0)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sampled", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    for name, (N, A, E, T, league) in {"PPO loop, 4096 x 4, episode 4096, horizon 64": (4096, 4, 4096, 64, False), "PPO loop, 4096 x 4, episode 64 = horizon": (4096, 4, 64, 64, False),
                                       "league, 2048 x 8, 2 trainable, episode 64 = horizon": (2048, 8, 64, 64, True)}.items():
        base = run(N, A, E, T, a.iters, 0, league)
        withr = run(N, A, E, T, a.iters, a.sampled, league)
        print(f"{name}: rollout {base['rollout_ms']:.3f} ms -> {withr['rollout_ms']:.3f} ms with {a.sampled} recorded markets ({100 * (withr['rollout_ms'] / base['rollout_ms'] - 1):+.1f} %); "
              f"update {base['update_ms']:.3f} -> {withr['update_ms']:.3f} ms; recorder host time per rollout {withr['host_record_ms']:.2f} ms (outside the device times: copies of the sampled "
              f"slices + Arrow columns at episode ends); {withr['rows']} rows written, NAV conservation checked on {withr['nav_checked']} episode ends: {withr['nav_violations']} violations")


if __name__ == "__main__":
    main()
