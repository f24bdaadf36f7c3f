#!/usr/bin/env python3
"""Search (with the CPU oracle, unbounded book) for market seeds whose book outgrows the product's LDS tile under the goldens'
"aggressive" law - the candidates tests/golden/make_goldens.py then cuts from the REAL reference (trace_bigbook_*).

    PYTHONPATH=tests/golden/shim:/root/reference:. python tests/golden/find_big_book_seed.py [agents steps n_seeds init_cash]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402
from make_goldens import sample_actions  # noqa: E402


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    cash = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
    law = sys.argv[5] if len(sys.argv) > 5 else "aggressive"
    cfg = {"num_of_agents": A, "init_cash": cash, "max_step": T, "is_render": False}
    env = O.OracleEnv(cfg, n_markets=S)
    env.set_book_cap(0)
    seeds = np.arange(200, 200 + S, dtype=np.uint64)
    env.reset(seeds=seeds)
    rngs = [np.random.default_rng(7200 + int(s)) for s in seeds]
    first = np.full(S, -1)
    for t in range(T):
        acts = [sample_actions(r, A, law, t) for r in rngs]
        cat, mean, sigma, price, off = (np.stack([a[k] for a in acts]) for k in range(5))
        env.step(cat, mean, sigma, price, off)
        if t % 64 == 63 or t == T - 1:
            pk = env.book_peak()
            first[(first < 0) & (pk > 512)] = t
    pk = env.book_peak()
    order = np.argsort(-pk)
    for i in order[:10]:
        print(f"seed {int(seeds[i])} action_seed {7200 + int(seeds[i])}: peak {int(pk[i])} first>512 by step {int(first[i])} now {env.book_size(int(i))}")
    print("markets over 512:", int((pk > 512).sum()), "of", S)


if __name__ == "__main__":
    main()
