"""Book census (run on the GPU box): the most resting orders any market holds, per agent count and action law.

    python tools/book_census.py [--markets 4096 --steps 4096] > profiles/r03/book_census.json

Laws: "uniform" = the RandomRLModule law of the reference (train/model/model_handler.py:38-53), every action component
uniform over its space; "aggressive" = the flip-heavy law of tests/golden/make_goldens.py (crossing limit / market
orders near the touch, small sizes); "trend" = the drifting market of the big-book goldens (orders are left behind by the
thousand).  The product keeps the top of a book in its LDS tile (256 / 512 orders) and the rest in its HBM spill ring; the
reference's OrderTree is unbounded.

(The round-2 census reported books "growing without bound" under the aggressive law at 16 agents.  That was an artefact of its
own loop: with groups = 2 it handed freshly allocated action tensors to group streams that were not ordered after the
caller's stream - ADVICE r2 - so the kernels read half-written actions.  CDAVecEnv.step now forks / joins by default.)
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402


def aggressive(gen, n, a, dev):
    cat_t = torch.tensor([1, 2, 2, 5, 6, 6, 3, 7, 4, 8], dtype=torch.int32, device=dev)
    off_t = torch.tensor([1, 2, 2], dtype=torch.int32, device=dev)
    cat = cat_t[torch.randint(0, 10, (n, a), generator=gen, device=dev)]
    price = torch.randint(0, 3, (n, a), generator=gen, device=dev, dtype=torch.int32)
    off = off_t[torch.randint(0, 3, (n, a), generator=gen, device=dev)]
    mean = (torch.rand((n, a), generator=gen, device=dev) * 0.1 - 0.05).float()
    sigma = torch.rand((n, a), generator=gen, device=dev).float()
    return cat, mean, sigma, price, off


def trend(gen, n, a, dev):
    cat_t = torch.tensor([2, 2, 2, 2, 2, 6, 6, 6, 1, 1, 1, 5, 3, 7, 4, 8, 0], dtype=torch.int32, device=dev)
    price_t = torch.tensor([0, 0, 0, 1, 2, 5, 9], dtype=torch.int32, device=dev)
    off_t = torch.tensor([2, 2, 1, 0], dtype=torch.int32, device=dev)
    cat = cat_t[torch.randint(0, 17, (n, a), generator=gen, device=dev)]
    price = price_t[torch.randint(0, 7, (n, a), generator=gen, device=dev)]
    off = off_t[torch.randint(0, 4, (n, a), generator=gen, device=dev)]
    mean = (torch.rand((n, a), generator=gen, device=dev) * 0.008 - 0.004).float()
    sigma = torch.rand((n, a), generator=gen, device=dev).float()
    return cat, mean, sigma, price, off


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--markets", type=int, default=4096)
    p.add_argument("--steps", type=int, default=4096)
    args = p.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for a in (4, 8, 16):
        for law in ("uniform", "aggressive", "trend"):
            cfg = {"num_of_agents": a, "init_cash": 1000000, "max_step": args.steps + 1, "is_render": False}
            steps = args.steps if law != "trend" else min(args.steps, 1024)
            cfg["max_step"] = steps + 1
            env = CDAVecEnv(cfg, args.markets, with_info=False, groups=2)
            env.reset(seed=1000)
            if law == "uniform":                            # whole episodes in one launch (cda_run_random)
                env.run_random(steps, action_seed=2024)
            else:
                gen = torch.Generator(device=dev)
                gen.manual_seed(7)
                law_fn = aggressive if law == "aggressive" else trend
                for _ in range(steps):
                    env.step(*law_fn(gen, args.markets, a, dev))
            peak, flags = env.book_peak(), env.flags()
            q = torch.quantile(peak.float(), torch.tensor([0.5, 0.99, 0.999], device=dev)).tolist()
            rows.append({"agents": a, "law": law, "markets": args.markets, "steps": steps, "tile": env.book_capacity, "spill_per_side": env.book_spill,
                         "max_resting_orders": int(peak.max()),
                         "p50": q[0], "p99": q[1], "p99.9": q[2], "markets_beyond_tile": int((peak > env.book_capacity).sum()),
                         "overflow_flagged_markets": int(((flags & 1) != 0).sum()), "other_flagged_markets": int(((flags & ~1) != 0).sum()),
                         "invariant_violations": int((env.check_invariants() != 0).sum())})
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
            env.close()
    print(json.dumps({"census": rows}, indent=1))


if __name__ == "__main__":
    main()
