"""How much of the per-step barrier does concurrency between market GROUPS buy back?  (run on the GPU box)

The batch is cut into G groups of N/G markets, each a chain of k_step launches on its own stream; the chains are
independent (markets never interact), so one group's straggler tail overlaps the other groups' bodies.  Modes:
direct launches (host enqueue cost included) and one captured HIP graph of T steps x G streams (GPU-side limit).

    python tools/groups_probe.py [--markets 4096 --agents 4 --steps 512]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402


def actions(n, a, steps, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cat = torch.randint(0, 9, (steps, n, a), generator=g, device=dev, dtype=torch.int32)
    price = torch.randint(0, 10, (steps, n, a), generator=g, device=dev, dtype=torch.int32)
    off = torch.randint(0, 3, (steps, n, a), generator=g, device=dev, dtype=torch.int32)
    mean = torch.rand((steps, n, a), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0
    sigma = torch.rand((steps, n, a), generator=g, device=dev, dtype=torch.float32)
    return cat, mean, sigma, price, off


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--markets", type=int, default=4096)
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--steps", type=int, default=512)
    p.add_argument("--groups", type=str, default="1,2,3,4,8")
    p.add_argument("--info", action="store_true")
    args = p.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    N, A, K = args.markets, args.agents, args.steps
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}
    CH = 64
    out = []
    for G in [int(x) for x in args.groups.split(",")]:
        per = [(N * (g + 1)) // G - (N * g) // G for g in range(G)]
        envs = [CDAVecEnv(cfg, n_markets=n, device="cuda:0", with_info=args.info) for n in per]
        first = 0
        for e, n in zip(envs, per):
            e.reset(seed=(1000 + first + torch.arange(n, dtype=torch.int64)))
            first += n
        acts = [actions(n, A, CH, dev, 2024 + g) for g, n in enumerate(per)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
        torch.cuda.synchronize()

        def enqueue(t0, steps):
            for t in range(t0, t0 + steps):
                i = t % CH
                for g in range(G):
                    with torch.cuda.stream(streams[g]):
                        a = acts[g]
                        envs[g].step(a[0][i], a[1][i], a[2][i], a[3][i], a[4][i])

        enqueue(0, 64)                               # warm the books up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueue(64, K)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_direct = time.perf_counter() - t0
        # one graph of CH steps x G chains
        main_s = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        t_graph = None
        try:
            with torch.cuda.stream(main_s):
                graph.capture_begin()
                for s in streams:
                    s.wait_stream(main_s)
                enqueue(0, CH)
                for s in streams:
                    main_s.wait_stream(s)
                graph.capture_end()
            torch.cuda.synchronize()
            reps = max(1, K // CH)
            with torch.cuda.stream(main_s):
                graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    graph.replay()
                torch.cuda.synchronize()
                t_graph = (time.perf_counter() - t0) / (reps * CH)
        except Exception as ex:  # noqa: BLE001
            print(f"G={G}: graph capture failed: {ex}", file=sys.stderr)
        flagged = sum(int((e.flags() != 0).sum().item()) for e in envs)
        row = {"groups": G, "markets": N, "agents": A, "info": args.info,
               "direct_us_per_step": t_direct / K * 1e6, "host_enqueue_us_per_step": t_enq / K * 1e6,
               "direct_Msteps": N * A * K / t_direct / 1e6,
               "graph_us_per_step": None if t_graph is None else t_graph * 1e6,
               "graph_Msteps": None if t_graph is None else N * A / t_graph / 1e6, "flagged": flagged}
        print(json.dumps(row), flush=True)
        out.append(row)
        for e in envs:
            e.close()
        del graph


if __name__ == "__main__":
    main()
