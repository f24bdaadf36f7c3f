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
    law = str(rng.choice(["uniform", "uniform", "aggressive", "edges", "trend"]))
    present_p = None if rng.random() < 0.7 else float(rng.uniform(0.3, 0.9))
    if rng.random() < 0.3:                                    # an integer tick other than 1 (round 6): the ladder's step and the unit of the observation's spread
        cfg["tick_size"] = int(rng.choice([2, 3, 5, 10, 250]))
    return cfg, law, present_p


def random_order(rng):
    """how the caller's action dicts are keyed: None = ascending agent ids, "shuffle" = a new random key order every step"""
    return "shuffle" if rng.random() < 0.35 else None


def prefill_book(env, market, rng, agents, n_bids, n_asks):
    """Give `market` of `env` (product or oracle: same get_state / set_state) a deep book through the state dump: n_bids / n_asks
    resting orders (up to 512 per side, far more than the LDS tile) in levels of one to four orders around the market's price,
    with the traders' escrow (cash_on_hold) set to match.  Both envs get the same book when called with equally seeded `rng`s."""
    from decimal import Decimal

    from gym_continuousdoubleauction_amd import _capi as K
    s = env.get_state(market)
    lp = max(2, int(s.last_price))
    hold = [0] * agents
    oid = 0
    for side, n, arr in ((0, n_bids, s.bids), (1, n_asks, s.asks)):
        price, left_in_level = (lp - 1 if side == 0 else lp + 1), int(rng.integers(1, 5))
        for k in range(n):
            if left_in_level == 0:
                step = int(rng.integers(1, 3))
                price = max(1, price - step) if side == 0 else price + step
                left_in_level = int(rng.integers(1, 5))
            left_in_level -= 1
            o = arr[k]
            oid += 1
            o.price, o.qty, o.owner, o.order_id, o.timestamp = price, int(rng.integers(1, 6)), int(rng.integers(0, agents)), oid, oid
            hold[o.owner] += o.price * o.qty
    s.n_bids, s.n_asks = n_bids, n_asks
    s.lob_time = s.next_order_id = oid
    for j in range(agents):
        cash = K.dec_to_decimal(s.acc[j].cash)
        s.acc[j].cash_on_hold = K.decimal_to_dec(Decimal(hold[j]) * Decimal("1.0"))
        s.acc[j].cash = K.decimal_to_dec(cash - Decimal(hold[j]) * Decimal("1.0"))
    env.set_state(market, s)


def batch_actions(rng, n, a, law, present_p, order=None):
    """[n, a] action arrays under one of the golden generator's laws, plus `present` (or None): a 0 / 1 mask, or - order ==
    "shuffle" - each market's own random dict order (0 = absent, else 1 + position in the dict; cda_step's encoding)."""
    if law == "uniform":
        cat, price, off = rng.integers(0, 9, (n, a)), rng.integers(0, 10, (n, a)), rng.integers(0, 3, (n, a))
        mean, sigma = rng.uniform(-1, 1, (n, a)), rng.uniform(0, 1, (n, a))
    elif law == "aggressive":
        cat, price, off = rng.choice([1, 2, 2, 5, 6, 6, 3, 7, 4, 8], (n, a)), rng.integers(0, 3, (n, a)), rng.choice([1, 2, 2], (n, a))
        mean, sigma = rng.uniform(-0.05, 0.05, (n, a)), rng.uniform(0, 1, (n, a))
    elif law == "trend":
        cat = rng.choice([2, 2, 2, 2, 2, 6, 6, 6, 1, 1, 1, 5, 3, 7, 4, 8, 0], (n, a))
        price, off = rng.choice([0, 0, 0, 1, 2, 5, 9], (n, a)), rng.choice([2, 2, 1, 0], (n, a))
        mean, sigma = rng.uniform(-0.004, 0.004, (n, a)), rng.uniform(0, 1, (n, a))
    else:
        cat, price, off = rng.integers(0, 9, (n, a)), rng.choice([0, 9], (n, a)), rng.choice([0, 2], (n, a))
        mean, sigma = rng.choice([-1.0, 1.0, 0.0], (n, a)), rng.choice([0.0, 1.0], (n, a))
    present = None if present_p is None else (rng.uniform(0, 1, (n, a)) < present_p).astype(np.uint8)
    if order == "shuffle":
        mask = np.ones((n, a), np.uint8) if present is None else present
        keys = rng.random((n, a)) + (mask == 0) * 2.0                      # absent agents sort last
        rank = np.argsort(np.argsort(keys, axis=1), axis=1)                 # 0-based position of each agent in its market's dict
        present = np.where(mask != 0, rank + 1, 0).astype(np.uint8)
    return (cat.astype(np.int32), mean.astype(np.float32), sigma.astype(np.float32), price.astype(np.int32), off.astype(np.int32)), present
