#!/usr/bin/env python3
"""How well do the rollout chains overlap?  Times RolloutChains.run() at 4096 x 4, horizon 64, for 1 / 2 / 4 / 8 chains, with and without
HIP graphs, and reports when each chain started and finished inside the rollout (HIP events on the chains' streams)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_continuousdoubleauction_amd import CDAVecEnv, mlp  # noqa: E402


def main():
    N, A, T = 4096, 4, 64
    dev = torch.device("cuda:0")
    out = []
    for groups, graphs, side in ((1, True, False), (4, True, False), (4, True, True), (4, False, True), (2, True, True), (8, True, True)):
        ctx = torch.cuda.stream(torch.cuda.Stream()) if side else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            env = CDAVecEnv({"num_of_agents": A, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}, n_markets=N, with_info=False)
            env.reset(seed=1)
            p = mlp.FusedPolicy(dev, seed=1)
            roll = mlp.RolloutChains(env, p, T, groups=groups, seed=3, use_graphs=graphs)
            for _ in range(3):
                roll.run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                roll.run()
                t_host = time.perf_counter() - t0
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0, t_host))
            # spans: events at the head and tail of every chain's stream
            ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
            heads, tails = [], []
            roll.counter.add_(1)
            for g, s in enumerate(roll.streams):
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(s)
                    if roll.graphs is not None:
                        roll.graphs[g].replay()
                    else:
                        roll._enqueue(g, True)
                    b.record(s)
                    heads.append(a); tails.append(b)
            torch.cuda.synchronize()
            spans = [(round(ev0.elapsed_time(a), 2), round(ev0.elapsed_time(b), 2)) for a, b in zip(heads, tails)]
            best = min(ts)
            r = {"chains": groups, "graphs": graphs, "caller_stream": "side" if side else "default", "rollout_ms": best[0] * 1e3, "host_enqueue_ms": best[1] * 1e3, "us_per_step": best[0] / T * 1e6,
                 "agent_steps_per_s": N * A * T / best[0], "chain_spans_ms": spans}
            print(json.dumps(r))
            out.append(r)
            env.close()
    return out


if __name__ == "__main__":
    main()
