"""Random env configurations and action laws shared by the build-container cross-check of the oracle against the reference
(tests/golden/crosscheck_oracle.py) and the GPU fuzz test of the HIP path against the oracle."""
import numpy as np


def random_config(rng):
    A = int(rng.choice([2, 3, 4, 5, 8, 12, 16]))
    cfg = {"num_of_agents": A, "init_cash": int(rng.choice([400, 3000, 50000, 1000000, 1000000, 50000000000])),
           "max_step": 4096, "is_render": False, "n_hist": int(rng.choice([1, 2, 4, 4, 6]))}
    if rng.random() < 0.4:
        lo = int(rng.choice([1, 10, 500, 20000]))
        cfg.update(initial_price_min=lo, initial_price_max=lo + int(rng.integers(0, 300)))
    if rng.random() < 0.4:
        cfg.update(min_size=int(rng.integers(1, 5)), mkt_max_size=int(rng.choice([20, 100, 3000])), limit_size_multiple=int(rng.choice([1, 3, 10, 20])))
    if rng.random() < 0.3:
        cfg.update(order_penalty=float(rng.uniform(0, 1)), trade_penalty=float(rng.uniform(0, 1)), drawdown_penalty=float(rng.uniform(0, 1)),
                   passive_bonus=float(rng.uniform(0, 1)), loss_multiplier=float(rng.uniform(1, 3)))
    law = str(rng.choice(["uniform", "uniform", "aggressive", "edges"]))
    present_p = None if rng.random() < 0.7 else float(rng.uniform(0.3, 0.9))
    return cfg, law, present_p


def batch_actions(rng, n, a, law, present_p):
    """[n, a] action arrays under one of the golden generator's laws, plus the present mask (or None)."""
    if law == "uniform":
        cat, price, off = rng.integers(0, 9, (n, a)), rng.integers(0, 10, (n, a)), rng.integers(0, 3, (n, a))
        mean, sigma = rng.uniform(-1, 1, (n, a)), rng.uniform(0, 1, (n, a))
    elif law == "aggressive":
        cat, price, off = rng.choice([1, 2, 2, 5, 6, 6, 3, 7, 4, 8], (n, a)), rng.integers(0, 3, (n, a)), rng.choice([1, 2, 2], (n, a))
        mean, sigma = rng.uniform(-0.05, 0.05, (n, a)), rng.uniform(0, 1, (n, a))
    else:
        cat, price, off = rng.integers(0, 9, (n, a)), rng.choice([0, 9], (n, a)), rng.choice([0, 2], (n, a))
        mean, sigma = rng.choice([-1.0, 1.0, 0.0], (n, a)), rng.choice([0.0, 1.0], (n, a))
    present = None if present_p is None else (rng.uniform(0, 1, (n, a)) < present_p).astype(np.uint8)
    return (cat.astype(np.int32), mean.astype(np.float32), sigma.astype(np.float32), price.astype(np.int32), off.astype(np.int32)), present
