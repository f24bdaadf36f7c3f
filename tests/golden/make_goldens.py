#!/usr/bin/env python3
"""Cut golden vectors from the REAL reference (imported read-only from /root/reference).

Run in the build container only:

    PYTHONPATH=tests/golden/shim:/root/reference python tests/golden/make_goldens.py

`gymnasium` and `ray` are absent from the image, so tests/golden/shim/ provides the ~100-line
stand-ins the reference's hard imports need.  Nothing of the reference is
copied: this script drives `continuousDoubleAuctionEnv` with seeded action streams and records
inputs, outputs and internal state as plain arrays (tests/golden/*.npz).  The fixtures are data;
the reference never travels to the GPU box.

Per trace and per step it records: the actions, obs (f32), rewards (f64), flags, the pre-step raw
snapshot, decoded orders, execution order, every account's Decimal triples, counters, info floats,
the book in queue order, LOB clocks and the numpy PCG64 state.
"""
import json
import os
import sys
from decimal import Decimal

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

from gym_continuousDoubleAuction.envs.continuousDoubleAuction_env import continuousDoubleAuctionEnv  # noqa: E402

ACC_FIELDS = ["cash", "cash_on_hold", "position_val", "VWAP", "nav", "prev_nav", "max_nav"]
TYPE_CODE = {"market": 0, "limit": 1, "modify": 2, "cancel": 3}
SIDE_CODE = {"bid": 0, "ask": 1, None: 2}


def dec_triple(d):
    assert isinstance(d, Decimal), type(d)
    sign, digits, exp = d.as_tuple()
    coeff = int("".join(map(str, digits)) or "0")
    assert coeff < 10 ** 28 and -32768 <= exp <= 32767
    return sign, coeff, exp


def sample_actions(rng, A, law, t=0):
    if law == "trend_waves":   # the "trend" law below, its direction reversed every 384 steps (sides swapped: 1 <-> 5, 2 <-> 6, ...):
        # the orders one wave leaves behind are swept, modified and cancelled by the next, far beyond the top of the book
        cat, mean, sigma, price, off = sample_actions(rng, A, "trend")
        if (t // 384) % 2 == 1:
            cat = np.where(cat == 0, 0, np.where(cat <= 4, cat + 4, cat - 4)).astype(np.int32)
        return cat, mean, sigma, price, off
    if law == "uniform":       # RandomRLModule / spaces.sample() law (train/model/model_handler.py:38-53)
        cat = rng.integers(0, 9, A)
        price = rng.integers(0, 10, A)
        off = rng.integers(0, 3, A)
        mean = rng.uniform(-1, 1, A).astype(np.float32)
        sigma = rng.uniform(0, 1, A).astype(np.float32)
    elif law == "aggressive":  # many crossing limit/market orders, small sizes -> many fills and flips
        cat = rng.choice([1, 2, 2, 5, 6, 6, 3, 7, 4, 8], A)
        price = rng.integers(0, 3, A)
        off = rng.choice([1, 2, 2], A)
        mean = (rng.uniform(-0.05, 0.05, A)).astype(np.float32)
        sigma = rng.uniform(0, 1, A).astype(np.float32)
    elif law == "edges":       # bounds of the action space
        cat = rng.integers(0, 9, A)
        price = rng.choice([0, 9], A)
        off = rng.choice([0, 2], A)
        mean = rng.choice(np.array([-1.0, 1.0, 0.0], np.float32), A)
        sigma = rng.choice(np.array([0.0, 1.0], np.float32), A)
    elif law == "trend":       # a drifting market that leaves resting orders behind: bids keep improving by a tick, market buys
        # lift the asks, few cancels - the book outgrows any fixed pool (hundreds of price levels, > 1000 resting orders)
        cat = rng.choice([2, 2, 2, 2, 2, 6, 6, 6, 1, 1, 1, 5, 3, 7, 4, 8, 0], A)
        price = rng.choice([0, 0, 0, 1, 2, 5, 9], A)
        off = rng.choice([2, 2, 1, 0], A)
        mean = (rng.uniform(-0.004, 0.004, A)).astype(np.float32)
        sigma = rng.uniform(0, 1, A).astype(np.float32)
    else:
        raise ValueError(law)
    return cat.astype(np.int32), mean, sigma, price.astype(np.int32), off.astype(np.int32)


def dump_side(tree):
    """Orders in queue order: best price first, FIFO inside a level."""
    rows = []
    items = list(tree.price_map.items())
    return items


def book_rows(lob):
    rows_b, rows_a = [], []
    for price, olist in reversed(list(lob.bids.price_map.items())):
        for o in olist:
            rows_b.append(o)
    for price, olist in lob.asks.price_map.items():
        for o in olist:
            rows_a.append(o)
    out = []
    for side_rows, tree in ((rows_b, lob.bids), (rows_a, lob.asks)):
        for o in side_rows:
            p = o.price
            assert p == p.to_integral_value() and p.as_tuple().exponent == -1, p
            q = o.quantity
            assert q == int(q)
            out.append((int(p), int(q), int(o.trade_id), int(o.order_id), int(o.timestamp)))
        # claim used by the build: among one trader's orders at one price, dict insertion
        # order of order_map equals FIFO order inside the level.
        seen = {}
        for oid, o in tree.order_map.items():
            seen.setdefault((o.trade_id, o.price), []).append(oid)
        fifo = {}
        for o in side_rows:
            fifo.setdefault((o.trade_id, o.price), []).append(o.order_id)
        assert seen == fifo, (seen, fifo)
        assert len(tree.order_map) == len(side_rows)
    return out, len(rows_b), len(rows_a)


def run_trace(name, config, seed, T, action_seed, law="uniform", present_p=None, reseed_at=None, presets=None, dict_order=None, book_every=1):
    """book_every > 1: the book dump (thousands of orders in the big-book traces) is kept for every book_every-th step and the
    last one only; `book_off[t] = (-1, n_bids, n_asks)` for the others (the counts are always recorded)."""
    env = continuousDoubleAuctionEnv(dict(config))
    A = env.num_of_agents
    obs0, _ = env.reset(seed=seed)
    od = env.n_hist * 42
    rec = {
        "config": np.array(json.dumps(config)), "seed": np.array(seed, np.uint64), "law": np.array(law),
        "obs0": obs0["agent_0"].copy(), "last_price0": np.array(env.last_price),
    }
    st = env.np_random.bit_generator.state
    rec["rng_inc"] = np.array([st["state"]["inc"] >> 64, st["state"]["inc"] & (2 ** 64 - 1)], np.uint64)
    rec["rng0"] = np.array([st["state"]["state"] >> 64, st["state"]["state"] & (2 ** 64 - 1), st["has_uint32"], st["uinteger"]], np.uint64)
    rng = np.random.default_rng(action_seed)
    keys_i = ["cat", "price", "off", "present", "term", "trunc", "lob_time", "next_order_id", "tape_len", "last_trade_price",
              "done_mask", "n_acts", "t_step"]
    L = {k: [] for k in keys_i}
    mean_l, sigma_l, obs_l, rew_l, raw_l = [], [], [], [], []
    dtype_l, dside_l, dsize_l, dprice_l, exec_l = [], [], [], [], []
    acc_sign, acc_exp, acc_coeff = [], [], []
    netpos_l, ntr_l, cnt_l, pass_l, terms_l, inff_l, mkt_l, rng_l = [], [], [], [], [], [], [], []
    book_all, book_off = [], []
    resets = []
    preset_rows = []
    for t in range(T):
        for (pt, tr, fields) in (presets or []):
            if pt == t:   # overwrite account fields directly, as the reference's own unit tests do (test_accounting.py:143-150)
                acc = env.traders[tr].acc
                for f, v in fields.items():
                    setattr(acc, f, int(v) if f == "net_position" else Decimal(v))
                acc.cal_nav()
                preset_rows.append((t, tr, int(fields.get("cash", -1)), int(fields.get("position_val", -1)), int(fields.get("VWAP", -1)),
                                    int(fields.get("net_position", 0))))
        if reseed_at and t in reseed_at:       # mid-trace reset: seed=None keeps the stream
            o, _ = env.reset(seed=reseed_at[t])
            resets.append((t, -1 if reseed_at[t] is None else reseed_at[t]))
            rec[f"reset_obs_{t}"] = o["agent_0"].copy()
        cat, mean, sigma, price, off = sample_actions(rng, A, law, t)
        present = np.ones(A, np.uint8) if present_p is None else (rng.uniform(0, 1, A) < present_p).astype(np.uint8)
        env.set_agg_LOB()
        raw_pre = np.asarray(env.agg_LOB_raw, np.float32).copy()
        actions = {}
        # dict_order: the reference assigns its RNG draws (and builds its arrival list) in the ITERATION order of this dict
        # (action_helper.py:145-172); a non-ascending order is recorded as given, in the reference's own agent ids
        if isinstance(dict_order, str):             # "shuffle": a different key order every step
            order_t = [int(x) for x in rng.permutation(A)]
        else:
            order_t = list(range(A)) if dict_order is None else list(dict_order)
        if dict_order is not None:                  # recorded as cda_step's `present` carries it: 1 + position in the dict, 0 = absent
            pos = 0
            for a in order_t:
                if present[a]:
                    pos += 1
                    present[a] = pos
        for a in order_t:
            if present[a]:
                actions[f"agent_{a}"] = {
                    "category": np.int64(cat[a]), "size_mean": np.array([mean[a]], np.float32),
                    "size_sigma": np.array([sigma[a]], np.float32), "price": np.int64(price[a]),
                    "price_offset": np.int64(off[a]),
                }
        obs, rewards, terms, truncs, infos = env.step(actions)
        # ---- record
        L["cat"].append(cat); L["price"].append(price); L["off"].append(off); L["present"].append(present)
        mean_l.append(mean); sigma_l.append(sigma); raw_l.append(raw_pre)
        ob = obs["agent_0"]
        for a in range(A):
            assert obs[f"agent_{a}"] is ob
        assert ob.dtype == np.float32 and ob.shape == (od,)
        obs_l.append(ob.copy())
        rew_l.append(np.array([rewards[f"agent_{a}"] for a in range(A)], np.float64))
        L["term"].append(int(terms["__all__"])); L["trunc"].append(int(truncs["__all__"]))
        for a in range(A):
            assert terms[f"agent_{a}"] is False and truncs[f"agent_{a}"] is False
        dt = np.full(A, -9, np.int32); ds = np.full(A, -9, np.int32); dz = np.full(A, -9, np.int32); dp = np.full(A, -9, np.int32)
        for act in env.LOB_actions:
            a = int(act["ID"].split("_")[1])
            dt[a] = TYPE_CODE[act["type"]]; ds[a] = SIDE_CODE[act["side"]]; dz[a] = act["size"]
            pr = act["price"]; assert float(pr) == int(pr)
            dp[a] = int(pr)
        ex = np.full(A, -1, np.int32)
        for i, act in enumerate(env.shuffled_actions):
            ex[i] = int(act["ID"].split("_")[1])
        L["n_acts"].append(len(env.shuffled_actions))
        dtype_l.append(dt); dside_l.append(ds); dsize_l.append(dz); dprice_l.append(dp); exec_l.append(ex)
        sg = np.zeros((A, 7), np.uint8); ex_ = np.zeros((A, 7), np.int16); co = np.zeros((A, 7, 3), np.uint32)
        npos = np.zeros(A, np.int32); ntr = np.zeros(A, np.int32); cnt = np.zeros((A, 4), np.int32)
        ps = np.zeros(A, np.uint8); tm = np.zeros((A, 5), np.float64); inf = np.zeros((A, 6), np.float64)
        for a in range(A):
            acc = env.traders[a].acc
            info = infos[f"agent_{a}"]
            for j, f in enumerate(ACC_FIELDS):
                s, c, e = dec_triple(getattr(acc, f))
                sg[a, j] = s; ex_[a, j] = e
                co[a, j] = (c & 0xFFFFFFFF, (c >> 32) & 0xFFFFFFFF, (c >> 64) & 0xFFFFFFFF)
            assert info["NAV"] == str(acc.nav)
            assert info["reward"] == rewards[f"agent_{a}"]
            assert isinstance(acc.net_position, int)
            npos[a] = acc.net_position; ntr[a] = acc.num_trades
            assert info["net_position"] == acc.net_position and info["num_trades"] == acc.num_trades
            cnt[a] = (info["num_trades_step"], info["num_passive_fills_step"], info["order_step_placed"], info["num_rejected_step"])
            ps[a] = int(info["is_pass_action"])
            rt = info["reward_terms"]
            tm[a] = (rt["nav_term"], rt["order_penalty"], rt["trade_penalty"], rt["drawdown_penalty"], rt["passive_bonus"])
            inf[a] = (info["VWAP"], info["cash"], info["cash_on_hold"], info["position_val"], info["drawdown"], info["max_nav"])
        i0 = infos["agent_0"]
        nan = float("nan")
        mkt_l.append(np.array([i0["last_price"], nan if i0["best_bid"] is None else i0["best_bid"],
                               nan if i0["best_ask"] is None else i0["best_ask"],
                               nan if i0["spread"] is None else i0["spread"]], np.float64))
        assert env.last_price == float(int(env.last_price))
        acc_sign.append(sg); acc_exp.append(ex_); acc_coeff.append(co)
        netpos_l.append(npos); ntr_l.append(ntr); cnt_l.append(cnt); pass_l.append(ps); terms_l.append(tm); inff_l.append(inf)
        lob = env.LOB
        L["lob_time"].append(lob.time); L["next_order_id"].append(lob.next_order_id); L["tape_len"].append(len(lob.tape))
        L["last_trade_price"].append(int(lob.tape[-1]["price"]) if len(lob.tape) else 0)
        dm = 0
        for aid in env.done_set:
            dm |= 1 << int(aid.split("_")[1])
        L["done_mask"].append(dm); L["t_step"].append(env.t_step)
        rows, nb, na = book_rows(lob)
        if t % book_every == 0 or t == T - 1:
            book_off.append((len(book_all), nb, na)); book_all.extend(rows)
        else:
            book_off.append((-1, nb, na))
        st = env.np_random.bit_generator.state
        rng_l.append(np.array([st["state"]["state"] >> 64, st["state"]["state"] & (2 ** 64 - 1), st["has_uint32"],
                               st["uinteger"] if st["has_uint32"] else 0], np.uint64))
    for k in keys_i:
        rec[k] = np.array(L[k], np.int32)
    rec.update(
        mean=np.array(mean_l, np.float32), sigma=np.array(sigma_l, np.float32), obs=np.array(obs_l, np.float32),
        reward=np.array(rew_l, np.float64), raw_pre=np.array(raw_l, np.float32),
        dec_type=np.array(dtype_l), dec_side=np.array(dside_l), dec_size=np.array(dsize_l), dec_price=np.array(dprice_l),
        exec_order=np.array(exec_l), acc_sign=np.array(acc_sign), acc_exp=np.array(acc_exp), acc_coeff=np.array(acc_coeff),
        net_position=np.array(netpos_l), num_trades=np.array(ntr_l), counters=np.array(cnt_l), is_pass=np.array(pass_l),
        reward_terms=np.array(terms_l), info_floats=np.array(inff_l), market=np.array(mkt_l), rng=np.array(rng_l),
        book=np.array(book_all, np.int32).reshape(-1, 5), book_off=np.array(book_off, np.int64),
        resets=np.array(resets, np.int64).reshape(-1, 2),
        presets=np.array(preset_rows, np.int64).reshape(-1, 6),
    )
    if dict_order is not None and not isinstance(dict_order, str):
        assert sorted(dict_order) == list(range(A))
        rec["dict_order"] = np.array(dict_order, np.int32)
    if dict_order is not None:
        rec["ordered"] = np.array(1)                # `present` holds ranks, not a 0 / 1 mask
    nonint = sum(1 for s in acc_exp for v in s[:, 4] if v < -1)
    print(f"{name}: T={T} A={A} tape={L['tape_len'][-1]} max_orders={max(b[1] + b[2] for b in book_off)} "
          f"deep-exp NAVs={nonint} term={sum(L['term'])} rejected={int(np.array(cnt_l)[:, :, 3].sum())}")
    return rec


def main():
    out_dir = HERE
    base4 = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 256, "is_render": False}
    base8 = {"num_of_agents": 8, "init_cash": 1000000, "max_step": 256, "is_render": False}
    traces = {}
    only = set(sys.argv[1:])              # optional: regenerate just the named traces

    def add(name, *a, **k):
        if not only or name in only:
            traces[name] = run_trace(name, *a, **k)
    base16 = {"num_of_agents": 16, "init_cash": 1000000, "max_step": 256, "is_render": False}
    for s in range(8):
        add(f"A4_s{s}", base4, s, 256, 5000 + s)
    for s in range(8):
        add(f"A8_s{s}", base8, s, 256, 6000 + s)
    add("default_s11", {"is_render": False}, 11, 80, 7011)
    add("lowcash_s21", dict(base4, init_cash=3000), 21, 256, 7021)
    add("lowcash_s22", dict(base4, init_cash=400), 22, 256, 7022, law="aggressive")
    add("aggr_s23", dict(base4), 23, 256, 7023, law="aggressive")
    add("aggr8_s24", dict(base8), 24, 256, 7024, law="aggressive")
    add("edges_s25", dict(base4), 25, 192, 7025, law="edges")
    add("subset_s31", dict(base4), 31, 160, 7031, present_p=0.6)
    add("hist1_s41", dict(base4, n_hist=1), 41, 48, 7041)
    add("hist6_s42", dict(base4, n_hist=6), 42, 48, 7042)
    add("coef_s43", dict(base4, order_penalty=0.3, trade_penalty=0.07, drawdown_penalty=0.11,
                                                    passive_bonus=0.9, loss_multiplier=2.25, initial_price_min=500,
                                                    initial_price_max=5000, min_size=2, mkt_max_size=40,
                                                    limit_size_multiple=3), 43, 128, 7043)
    add("reset_s51", dict(base4, max_step=40), 51, 120, 7051, reseed_at={40: None, 80: 977})
    add("big_seed", dict(base4), 2 ** 63 + 12345, 64, 7061)
    # bankruptcies: heavily short accounts are marked against a much higher price -> NAV <= 0 -> done_set, rejections,
    # and finally terminateds["__all__"]
    short = lambda v: {"cash": 100, "position_val": 10 * v, "VWAP": 10, "net_position": -v}   # noqa: E731
    add("bankrupt_s61", dict(base4, max_step=96), 61, 96, 7161,
                                       presets=[(3, 0, short(60000)), (20, 1, short(80000)), (40, 2, short(90000)), (60, 3, short(70000))])
    # the build's agent-count bound (CDA_MAX_AGENTS = 16): owner lanes 0-15, helper lanes 16-47 all busy
    add("A16_s70", base16, 70, 160, 7070)
    add("A16_aggr_s71", dict(base16, init_cash=200000), 71, 128, 7071, law="aggressive")
    # large sizes and balances (11-digit cash, positions in the 10^5 range: longer coefficients in every ledger product) and
    # prices at the tick floor (the `price < tick -> tick` clamp, one-tick books)
    add("bigsize_s81", dict(base4, init_cash=50000000000, mkt_max_size=5000, limit_size_multiple=20, min_size=3), 81, 192, 7081)
    add("bigsize_aggr_s82", dict(base8, init_cash=50000000000, mkt_max_size=3000, limit_size_multiple=7), 82, 160, 7082, law="aggressive")
    add("floor_s83", dict(base4, initial_price_min=1, initial_price_max=3), 83, 192, 7083)
    add("long_s100", dict(base4, max_step=2048), 100, 2048, 7100)
    # action dicts handed over in a fixed NON-ascending key order (all agents, and subsets of them)
    add("perm_s91", dict(base4), 91, 160, 7091, dict_order=[2, 0, 3, 1])
    add("perm8_s92", dict(base8), 92, 128, 7092, law="aggressive", present_p=0.7, dict_order=[5, 1, 7, 0, 3, 6, 2, 4])
    # ... and in an order that changes every step (all agents / subsets, 4 and 8 agents)
    add("permshuf_s93", dict(base4), 93, 160, 7093, dict_order="shuffle")
    add("permshuf8_s94", dict(base8), 94, 128, 7094, law="aggressive", present_p=0.75, dict_order="shuffle")
    # books far beyond any fixed pool (the reference's OrderTree is unbounded, ordertree.py:5-58): a drifting market leaves
    # thousands of resting orders behind; in the second trace the drift reverses every 384 steps, so sweeps, modifies and
    # cancels reach deep into what the previous wave left.  (seeds found with tests/golden/find_big_book_seed.py)
    add("bigbook_s201", dict(base16, max_step=1280), 201, 1280, 7401, law="trend", book_every=64)                 # ~3000 resting bids (tile: 512)
    add("bigbook_waves_s205", dict(base16, max_step=1280), 205, 1280, 7405, law="trend_waves", book_every=64)     # ~850 orders, both sides deep
    add("bigbook8_waves_s203", dict(base8, max_step=1280), 203, 1280, 7403, law="trend_waves", book_every=64)     # tile 256
    add("bigbook4_s201", dict(base4, max_step=1280), 201, 1280, 7401, law="trend", book_every=64)                 # tile 256, 4 agents
    # integer tick sizes other than 1 (round 6): the price ladder steps in ticks (action_helper.py:341-397), the observation's spread is counted in them
    # (state_helper.py:202-206: a NON-integer argument of log1p once a price was clamped off the ladder's grid - the floor traces)
    add("tick5_s301", dict(base4, tick_size=5), 301, 192, 7301)
    add("tick3_floor_s302", dict(base4, tick_size=3, initial_price_min=1, initial_price_max=7), 302, 192, 7302)
    add("tick7_aggr8_s303", dict(base8, tick_size=7, initial_price_min=20, initial_price_max=60), 303, 160, 7303, law="aggressive")
    add("tick250_edges_s304", dict(base4, tick_size=250, initial_price_min=100, initial_price_max=3000), 304, 160, 7304, law="edges")
    for name, rec in traces.items():
        np.savez_compressed(os.path.join(out_dir, f"trace_{name}.npz"), **rec)
    if only:
        return 0
    tot = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir) if f.endswith(".npz"))
    print("total fixture bytes", tot)


if __name__ == "__main__":
    sys.exit(main())
