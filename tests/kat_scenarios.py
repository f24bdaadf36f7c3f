"""Known-answer scenarios restated (as data) from the reference's own hot-path unit tests:
  test/test_accounting.py, test_cash_check.py, test_modify_order.py, test_orderbook_new.py,
  test_orderbook_crossed_book.py, test_orderbook_volume_sync.py.
Each scenario is a list of ops driven through `place_order` (Trader.place_order, agent/trader.py:49)
on market 0 of ANY env adapter (the CPU oracle or the HIP product), with exact expectations.

ops:
  ("order", trader, type, side, size, price)   type: market/limit/modify/cancel; price None for market
  ("set", trader, {field: int})                preset account fields (cash, position_val, vwap, net_position)
  ("mtm", price)                               Calculate.mark_to_mkt at `price`
  ("acc", trader, field, value)                exact Decimal value of an account field; field "calc_nav"
                                               = cash + cash_on_hold + position_val (acc.cal_nav())
  ("book", {...})                              n_bids, n_asks, best_bid, best_ask, bid_volume, ask_volume,
                                               lob_time, next_order_id, tape (has_trade), last_trade_price
"""
from decimal import Decimal

from gym_continuousdoubleauction_amd import _capi as K

TYPE = {"market": 0, "limit": 1, "modify": 2, "cancel": 3}
SIDE = {"bid": 0, "ask": 1}

O = lambda *a: ("order",) + a      # noqa: E731

SCENARIOS = [
    # ---------------- test_accounting.py ----------------
    dict(name="acct_limit_placement_hold", cash=1000, ops=[
        O(0, "limit", "bid", 1, 100), ("acc", 0, "cash", 900), ("acc", 0, "cash_on_hold", 100), ("acc", 0, "nav", 1000),
        O(1, "limit", "ask", 1, 102), ("acc", 1, "cash", 898), ("acc", 1, "cash_on_hold", 102), ("acc", 1, "nav", 1000)]),
    dict(name="acct_limit_cancel", cash=1000, ops=[
        O(0, "limit", "bid", 1, 100), O(0, "cancel", "bid", 1, 100),
        ("acc", 0, "cash", 1000), ("acc", 0, "cash_on_hold", 0), ("acc", 0, "calc_nav", 1000),
        O(1, "limit", "ask", 1, 100), O(1, "cancel", "ask", 1, 100),
        ("acc", 1, "cash", 1000), ("acc", 1, "cash_on_hold", 0), ("acc", 1, "calc_nav", 1000), ("book", dict(n_bids=0, n_asks=0))]),
    dict(name="acct_market_short_matching", cash=1000, ops=[
        O(0, "limit", "bid", 1, 100), O(1, "market", "ask", 1, None),
        ("acc", 0, "cash_on_hold", 0), ("acc", 0, "position_val", 100), ("acc", 0, "net_position", 1), ("acc", 0, "cash", 900),
        ("acc", 0, "calc_nav", 1000),
        ("acc", 1, "cash", 900), ("acc", 1, "position_val", 100), ("acc", 1, "net_position", -1), ("acc", 1, "calc_nav", 1000)]),
    dict(name="acct_market_long_matching", cash=1000, ops=[
        O(0, "limit", "ask", 1, 100), O(1, "market", "bid", 1, None),
        ("acc", 0, "cash_on_hold", 0), ("acc", 0, "position_val", 100), ("acc", 0, "net_position", -1), ("acc", 0, "calc_nav", 1000),
        ("acc", 1, "cash", 900), ("acc", 1, "position_val", 100), ("acc", 1, "net_position", 1), ("acc", 1, "calc_nav", 1000)]),
    dict(name="acct_partial_fill", cash=1000, ops=[
        O(0, "limit", "bid", 2, 100), O(1, "market", "ask", 1, None),
        ("acc", 0, "cash", 800), ("acc", 0, "cash_on_hold", 100), ("acc", 0, "position_val", 100), ("acc", 0, "net_position", 1),
        ("acc", 0, "calc_nav", 1000), ("book", dict(n_bids=1, bid_volume=1))]),
    dict(name="acct_mtm_long", cash=1000, ops=[
        ("set", 0, dict(cash=900, position_val=100, net_position=1, vwap=100)),
        ("mtm", 110), ("acc", 0, "nav", 1010), ("mtm", 90), ("acc", 0, "nav", 990)]),
    dict(name="acct_mtm_short", cash=1000, ops=[
        ("set", 1, dict(cash=900, position_val=100, net_position=-1, vwap=100)),
        ("mtm", 110), ("acc", 1, "nav", 990), ("mtm", 90), ("acc", 1, "nav", 1010)]),
    dict(name="acct_market_order_empty_book", cash=1000, ops=[
        O(0, "market", "bid", 1, None), ("acc", 0, "cash", 1000), ("acc", 0, "calc_nav", 1000),
        ("book", dict(n_bids=0, n_asks=0, tape=0, lob_time=1, next_order_id=1))]),
    dict(name="acct_flip_long_to_short_aggressor", cash=1000, ops=[
        ("set", 0, dict(cash=900, net_position=1, position_val=100, vwap=100)),
        O(1, "limit", "bid", 2, 100), O(0, "market", "ask", 2, None),
        ("acc", 0, "net_position", -1), ("acc", 0, "position_val", 100), ("acc", 0, "cash", 900), ("acc", 0, "calc_nav", 1000)]),
    dict(name="acct_flip_short_to_long_aggressor", cash=1000, ops=[
        ("set", 0, dict(cash=900, net_position=-1, position_val=100, vwap=100)),
        O(1, "limit", "ask", 2, 100), O(0, "market", "bid", 2, None),
        ("acc", 0, "net_position", 1), ("acc", 0, "position_val", 100), ("acc", 0, "cash", 900), ("acc", 0, "calc_nav", 1000)]),
    dict(name="acct_flip_long_to_short_passive", cash=1000, ops=[
        ("set", 0, dict(cash=900, net_position=1, position_val=100, vwap=100)),
        O(0, "limit", "ask", 2, 100), ("acc", 0, "cash", 700), ("acc", 0, "cash_on_hold", 200),
        O(1, "limit", "bid", 2, 100),
        ("acc", 0, "net_position", -1), ("acc", 0, "position_val", 100), ("acc", 0, "cash", 900), ("acc", 0, "calc_nav", 1000)]),
    dict(name="acct_flip_short_to_long_passive", cash=1000, ops=[
        ("set", 0, dict(cash=900, net_position=-1, position_val=100, vwap=100)),
        O(0, "limit", "bid", 2, 100), ("acc", 0, "cash", 700), ("acc", 0, "cash_on_hold", 200),
        O(1, "limit", "ask", 2, 100),
        ("acc", 0, "net_position", 1), ("acc", 0, "position_val", 100), ("acc", 0, "cash", 900), ("acc", 0, "calc_nav", 1000)]),
    # ---------------- test_cash_check.py (trader 0 has 100, others preset) ----------------
    dict(name="cash_limit_buy_insufficient", cash=100, ops=[
        O(0, "limit", "bid", 1, 150), ("acc", 0, "cash", 100), ("acc", 0, "num_rejected_step", 1), ("book", dict(n_bids=0, lob_time=0))]),
    dict(name="cash_limit_buy_sufficient", cash=100, ops=[O(0, "limit", "bid", 1, 50), ("acc", 0, "cash", 50)]),
    dict(name="cash_market_buy_insufficient", cash=100, ops=[
        ("set", 1, dict(cash=1000)), O(1, "limit", "ask", 1, 200), O(0, "market", "bid", 1, None),
        ("acc", 0, "cash", 100), ("book", dict(tape=0, n_asks=1))]),
    dict(name="cash_cover_short_no_cash", cash=1000, ops=[
        ("set", 0, dict(cash=0, net_position=-1, position_val=100, vwap=100, nav=100)),
        O(1, "limit", "ask", 1, 100), O(0, "market", "bid", 1, None),
        ("acc", 0, "net_position", 0), ("acc", 0, "cash", 100), ("book", dict(tape=1))]),
    dict(name="cash_sell_long_no_cash", cash=1000, ops=[
        ("set", 0, dict(cash=0, net_position=1, position_val=100, vwap=100, nav=100)),
        O(0, "market", "ask", 1, None), ("acc", 0, "cash", 0),
        O(1, "limit", "bid", 1, 100), O(0, "market", "ask", 1, None), ("acc", 0, "cash", 100), ("book", dict(tape=1))]),
    dict(name="cash_flip_insufficient", cash=5000, ops=[
        ("set", 0, dict(cash=50, net_position=10, position_val=1000, vwap=100, nav=1050)),
        O(1, "limit", "bid", 20, 100), O(0, "market", "ask", 20, None),
        ("acc", 0, "cash", 50), ("acc", 0, "net_position", 10), ("book", dict(tape=0))]),
    dict(name="cash_estimate_uses_latest_tape_price", cash=5000, ops=[
        ("set", 0, dict(cash=1000)),
        O(1, "limit", "ask", 1, 100), O(0, "market", "bid", 1, None), ("book", dict(last_trade_price=100)),
        O(1, "limit", "ask", 1, 200), O(0, "market", "bid", 1, None), ("book", dict(last_trade_price=200, n_asks=0)),
        ("set", 0, dict(cash=150)), O(0, "market", "bid", 1, None), ("acc", 0, "num_trades", 2), ("acc", 0, "num_rejected_step", 1)]),
    # ---------------- test_modify_order.py (A = trader 0, B = trader 1) ----------------
    dict(name="modify_1_price_crosses_book", cash=10000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 10, 110),
        ("acc", 1, "cash", 9000), ("acc", 1, "cash_on_hold", 0), ("acc", 1, "net_position", 10)]),
    dict(name="modify_2_price_change_no_cross", cash=10000, ops=[
        O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 10, 95), ("acc", 1, "cash", 9050), ("acc", 1, "cash_on_hold", 950)]),
    dict(name="modify_3_qty_increase", cash=10000, ops=[
        O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 15, 90), ("acc", 1, "cash", 8650), ("acc", 1, "cash_on_hold", 1350)]),
    dict(name="modify_4_qty_decrease_same_price", cash=10000, ops=[
        O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 5, 90), ("acc", 1, "cash", 9550), ("acc", 1, "cash_on_hold", 450)]),
    dict(name="modify_5_cross_plus_qty_increase", cash=10000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 15, 110),
        ("acc", 1, "cash", 8450), ("acc", 1, "cash_on_hold", 550), ("acc", 1, "net_position", 10)]),
    dict(name="modify_6_cross_plus_qty_decrease", cash=10000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 5, 110),
        ("acc", 1, "cash", 9500), ("acc", 1, "cash_on_hold", 0), ("acc", 1, "net_position", 5)]),
    # ---------------- test_orderbook_new.py / crossed_book / volume_sync ----------------
    dict(name="lob_passive_rest", cash=100000, ops=[
        O(0, "limit", "bid", 10, 100), ("book", dict(n_bids=1, best_bid=100, bid_volume=10, tape=0))]),
    dict(name="lob_full_match", cash=100000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "bid", 10, 100), ("book", dict(n_asks=0, n_bids=0, ask_volume=0, tape=1))]),
    dict(name="lob_partial_match_residual", cash=100000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "bid", 15, 100), ("book", dict(bid_volume=5, best_bid=100, n_asks=0))]),
    dict(name="lob_market_sweeps_two_levels", cash=100000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "ask", 10, 101), O(2, "market", "bid", 15, None),
        ("book", dict(ask_volume=5, best_ask=101, n_asks=1)), ("acc", 2, "num_trades", 2)]),
    dict(name="lob_cancel", cash=100000, ops=[
        O(0, "limit", "bid", 10, 100), O(0, "cancel", "bid", 10, 100), ("book", dict(bid_volume=0, n_bids=0, lob_time=2))]),
    dict(name="lob_modify_qty_decrease", cash=100000, ops=[
        O(0, "limit", "bid", 10, 100), O(0, "modify", "bid", 5, 100), ("book", dict(bid_volume=5, n_bids=1))]),
    dict(name="lob_modify_price_change", cash=100000, ops=[
        O(0, "limit", "bid", 10, 100), O(0, "modify", "bid", 10, 101), ("book", dict(best_bid=101, n_bids=1, bid_volume=10))]),
    dict(name="lob_empty_book_market", cash=100000, ops=[O(0, "market", "bid", 10, None), ("book", dict(tape=0))]),
    dict(name="lob_order_ids_unique", cash=100000, ops=[
        O(0, "limit", "bid", 1, 100), O(1, "limit", "bid", 1, 100), ("book", dict(n_bids=2, order_ids_unique=True, next_order_id=2))]),
    dict(name="lob_modify_never_crosses", cash=100000, ops=[
        O(0, "limit", "ask", 10, 100), O(1, "limit", "bid", 10, 90), O(1, "modify", "bid", 10, 110),
        ("book", dict(n_bids=0, n_asks=0, tape=1, last_trade_price=100))]),
    dict(name="lob_volume_sync_after_partial_fill", cash=100000, ops=[
        O(0, "limit", "bid", 10, 100), O(1, "market", "ask", 4, None), ("book", dict(bid_volume=6, n_bids=1))]),
    # ---------------- extra: queue priority, upsert, self trade ----------------
    dict(name="x_time_priority_and_upsert", cash=100000, ops=[
        O(0, "limit", "bid", 5, 100), O(1, "limit", "bid", 7, 100), O(0, "limit", "bid", 9, 100),   # upsert qty up: loses priority
        O(2, "market", "ask", 7, None),
        ("acc", 1, "net_position", 7), ("acc", 0, "net_position", 0), ("book", dict(bid_volume=9, n_bids=1))]),
    dict(name="x_self_trade_moves_escrow_only", cash=100000, ops=[
        O(0, "limit", "bid", 5, 100), O(0, "market", "ask", 5, None),
        ("acc", 0, "cash", 100000), ("acc", 0, "cash_on_hold", 0), ("acc", 0, "net_position", 0), ("acc", 0, "num_trades", 0),
        ("book", dict(tape=1, n_bids=0))]),
]


def _dec_int(v):
    return K.decimal_to_dec(Decimal(int(v)))


def run_scenario(env, sc):
    """env: adapter with place_order / mark_to_mkt / get_state / set_state on market 0 (reset already done)."""
    for op in sc["ops"]:
        kind = op[0]
        if kind == "order":
            _, tr, typ, side, size, price = op
            env.place_order(0, tr, TYPE[typ], SIDE[side], size, 1 if price is None else price)
        elif kind == "set":
            s = env.get_state(0)
            acc = s.acc[op[1]]
            for f, v in op[2].items():
                if f == "net_position":
                    acc.net_position = int(v)
                else:
                    setattr(acc, f, _dec_int(v))
            if "nav" not in op[2]:      # the reference tests call cal_nav() after presetting
                tot = sum(K.dec_to_decimal(getattr(acc, f)) for f in ("cash", "cash_on_hold", "position_val"))
                acc.nav = K.decimal_to_dec(tot)
            env.set_state(0, s)
        elif kind == "mtm":
            s = env.get_state(0)
            s.has_trade = 1
            s.last_trade_price = int(op[1])
            env.set_state(0, s)
            env.mark_to_mkt(0)
        elif kind == "acc":
            _, tr, field, value = op
            acc = env.get_state(0).acc[tr]
            if field == "calc_nav":
                got = sum(K.dec_to_decimal(getattr(acc, f)) for f in ("cash", "cash_on_hold", "position_val"))
            elif field in ("net_position", "num_trades", "num_rejected_step", "num_trades_step"):
                got = Decimal(int(getattr(acc, field)))
            else:
                got = K.dec_to_decimal(getattr(acc, field))
            assert got == Decimal(value), f"{sc['name']}: trader {tr} {field}: got {got}, expected {value}"
        elif kind == "book":
            s = env.get_state(0)
            bids = [s.bids[i] for i in range(s.n_bids)]
            asks = [s.asks[i] for i in range(s.n_asks)]
            view = dict(n_bids=s.n_bids, n_asks=s.n_asks, best_bid=bids[0].price if bids else None,
                        best_ask=asks[0].price if asks else None, bid_volume=sum(o.qty for o in bids),
                        ask_volume=sum(o.qty for o in asks), lob_time=s.lob_time, next_order_id=s.next_order_id,
                        tape=s.has_trade, last_trade_price=s.last_trade_price,
                        order_ids_unique=len({o.order_id for o in bids + asks}) == len(bids + asks))
            for k, v in op[1].items():
                assert view[k] == v, f"{sc['name']}: book {k}: got {view[k]}, expected {v}"
            # invariants of test_orderbook_new.py:31-84: sorted, never crossed
            assert all(bids[i].price >= bids[i + 1].price for i in range(len(bids) - 1))
            assert all(asks[i].price <= asks[i + 1].price for i in range(len(asks) - 1))
            if bids and asks:
                assert bids[0].price < asks[0].price
        else:
            raise ValueError(kind)
