"""ctypes mirror of include/cda.h (struct layouts and constants).

Kept free of any library loading: pure struct definitions and value conversions, importable anywhere.
"""
import ctypes as C

import numpy as np

K_ROWS = 10
SNAPSHOT_DIM = 42
RAW_DIM = 40
MAX_HIST = 16
MAX_AGENTS = 16
BOOK_CAP = 256          # default LDS book tile (the top of a market's book, both sides together)
BOOK_CAP_MAX = 512      # the larger compiled tile; sizes the arrays of the fixed-size parity dump (MarketState)
MAX_GROUPS = 16
NUM_REWARD_TERMS = 5

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_NOMEM = -1, -2, -3, -4, -5

FLAG_BOOK_OVERFLOW, FLAG_INT_OVERFLOW, FLAG_DEC_DOMAIN, FLAG_NAV_CONSERVATION = 1, 2, 4, 8

# episode metrics (include/cda.h CDA_EM_*): columns of the per-module table / of the per-env row cda_episode_metrics_collect returns
EM_AGENT_FIELDS, EM_ENV_FIELDS, EM_MAX_MODULES = 32, 8, 32
EM_EPISODES, EM_AGENT_STEPS, EM_PASSES, EM_REJECTIONS, EM_PLACED, EM_TRADES, EM_PASSIVE, EM_TERM_SUM, EM_TERM_SQ = 0, 1, 2, 3, 4, 5, 6, 7, 12
EM_RETURN_SUM, EM_RETURN_SQ, EM_NAV_SUM, EM_NAV_MIN, EM_NAV_MAX, EM_DRAWDOWN_SUM, EM_ABS_POSITION_SUM, EM_NUM_TRADES_SUM = 17, 18, 19, 20, 21, 22, 23, 24
EM_MAKER_RATIO_SUM, EM_MAKER_RATIO_N, EM_MAKER_RATIO_MAX, EM_BANKRUPT = 25, 26, 27, 28
EM_ENV_EPISODES, EM_ENV_NAV_VIOLATIONS, EM_ENV_NAV_ERROR_SUM, EM_ENV_NAV_ERROR_MAX, EM_ENV_MAKER_MAX_SUM, EM_ENV_MAKER_MAX_N, EM_ENV_STEPS, EM_ENV_TERMINATED = range(8)

T_MARKET, T_LIMIT, T_MODIFY, T_CANCEL = 0, 1, 2, 3
S_BID, S_ASK = 0, 1


class Config(C.Structure):
    _fields_ = [
        ("num_agents", C.c_int32), ("max_step", C.c_int32), ("n_hist", C.c_int32), ("tick_size", C.c_int32),
        ("init_cash", C.c_int64),
        ("initial_price_min", C.c_int32), ("initial_price_max", C.c_int32),
        ("min_size", C.c_int32), ("mkt_max_size", C.c_int32), ("limit_size_multiple", C.c_int32),
        ("auto_reset", C.c_int32), ("book_capacity", C.c_int32), ("book_spill", C.c_int32),
        ("order_penalty", C.c_double), ("trade_penalty", C.c_double), ("drawdown_penalty", C.c_double),
        ("passive_bonus", C.c_double), ("loss_multiplier", C.c_double),
    ]


class Dec(C.Structure):
    _fields_ = [("w", C.c_uint32 * 3), ("exp", C.c_int16), ("sign", C.c_uint8), ("pad", C.c_uint8)]


# numpy view of cda_dec (include/cda.h): 96-bit coefficient, int16 exponent, sign
DEC_DTYPE = np.dtype([("w", np.uint32, (3,)), ("exp", np.int16), ("sign", np.uint8), ("pad", np.uint8)])

INFO_FIELDS = [
    # name, element ctype, per-agent?, trailing dims
    ("nav", Dec, True, ()),
    ("num_trades", C.c_int32, True, ()),
    ("net_position", C.c_int32, True, ()),
    ("vwap", C.c_double, True, ()),
    ("cash", C.c_double, True, ()),
    ("cash_on_hold", C.c_double, True, ()),
    ("position_val", C.c_double, True, ()),
    ("drawdown", C.c_double, True, ()),
    ("max_nav", C.c_double, True, ()),
    ("num_trades_step", C.c_int32, True, ()),
    ("num_passive_fills_step", C.c_int32, True, ()),
    ("order_step_placed", C.c_int32, True, ()),
    ("num_rejected_step", C.c_int32, True, ()),
    ("is_pass_action", C.c_uint8, True, ()),
    ("reward_terms", C.c_double, True, (NUM_REWARD_TERMS,)),
    ("last_price", C.c_double, False, ()),
    ("best_bid", C.c_double, False, ()),
    ("best_ask", C.c_double, False, ()),
    ("spread", C.c_double, False, ()),
    ("lob_actions", C.c_int32, True, (4,)),
]


class InfoPtrs(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _t, _pa, _d in INFO_FIELDS]


class Order(C.Structure):
    _fields_ = [("price", C.c_int32), ("qty", C.c_int32), ("owner", C.c_int32),
                ("order_id", C.c_int32), ("timestamp", C.c_int32)]


class AccountState(C.Structure):
    _fields_ = [("cash", Dec), ("cash_on_hold", Dec), ("position_val", Dec), ("vwap", Dec),
                ("nav", Dec), ("prev_nav", Dec), ("max_nav", Dec),
                ("net_position", C.c_int32), ("num_trades", C.c_int32),
                ("num_trades_step", C.c_int32), ("num_passive_fills_step", C.c_int32),
                ("order_step_placed", C.c_int32), ("num_rejected_step", C.c_int32)]


class MarketState(C.Structure):
    _fields_ = [
        ("rng_state_hi", C.c_uint64), ("rng_state_lo", C.c_uint64),
        ("rng_inc_hi", C.c_uint64), ("rng_inc_lo", C.c_uint64),
        ("rng_has_uint32", C.c_uint32), ("rng_uinteger", C.c_uint32),
        ("t_step", C.c_int32), ("lob_time", C.c_int32), ("next_order_id", C.c_int32),
        ("last_price", C.c_int32), ("has_trade", C.c_int32), ("last_trade_price", C.c_int32),
        ("done_mask", C.c_uint32), ("flags", C.c_uint32),
        ("n_bids", C.c_int32), ("n_asks", C.c_int32),
        ("bids", Order * BOOK_CAP_MAX), ("asks", Order * BOOK_CAP_MAX),
        ("acc", AccountState * MAX_AGENTS),
        ("hist", C.c_float * (MAX_HIST * SNAPSHOT_DIM)),
    ]


# config/env_defaults.json:8-27 of the reference
ENV_DEFAULTS = {
    "num_of_agents": 5,
    "init_cash": 1000000,
    "tick_size": 1,
    "tape_display_length": 10,
    "max_step": 64,
    "is_render": True,
    "n_hist": 4,
    "initial_price_min": 10,
    "initial_price_max": 100,
    "min_size": 1,
    "mkt_max_size": 100,
    "limit_size_multiple": 10,
    "order_penalty": 0.1,
    "trade_penalty": 0.05,
    "drawdown_penalty": 0.2,
    "passive_bonus": 0.1,
    "loss_multiplier": 1.5,
}


def make_config(config=None):
    """Env-config dict (reference keys, continuousDoubleAuction_env.py:27-55) -> Config struct.

    Unknown keys raise; missing keys take the reference's standalone defaults."""
    cfg = dict(ENV_DEFAULTS)
    cfg["auto_reset"] = False            # extensions of this build (include/cda.h), not reference keys
    cfg["book_capacity"] = 0             # LDS book tile: 0 = by agent count (256 up to 8 agents, 512 above); or 256 / 512
    cfg["book_spill"] = 0                # HBM spill ring, orders per side: 0 = automatic (num_agents * max_step: never overflows inside an
    #                                      episode), n > 0 = at least n, -1 = no HBM tier (a rest beyond the tile is dropped and flagged)
    for k, v in (config or {}).items():
        if k not in cfg:
            raise KeyError(f"unknown env config key {k!r}; known: {sorted(cfg)}")
        cfg[k] = v
    init_cash = cfg["init_cash"]
    if int(init_cash) != init_cash:
        raise ValueError("init_cash must be integer valued")
    tick = cfg["tick_size"]
    if int(tick) != tick or not 1 <= int(tick) <= 65536:
        raise ValueError("tick_size must be an integer in 1 .. 65536 (off the integer grid the reference's float price arithmetic is platform dependent: SURVEY A.10)")
    c = Config()
    c.num_agents = int(cfg["num_of_agents"])
    c.max_step = int(cfg["max_step"])
    c.n_hist = int(cfg["n_hist"])
    c.tick_size = int(tick)
    c.init_cash = int(init_cash)
    c.initial_price_min = int(cfg["initial_price_min"])
    c.initial_price_max = int(cfg["initial_price_max"])
    c.min_size = int(cfg["min_size"])
    c.mkt_max_size = int(cfg["mkt_max_size"])
    c.limit_size_multiple = int(cfg["limit_size_multiple"])
    c.auto_reset = 1 if cfg["auto_reset"] else 0
    c.book_capacity = int(cfg["book_capacity"])
    c.book_spill = int(cfg["book_spill"])
    c.order_penalty = float(cfg["order_penalty"])
    c.trade_penalty = float(cfg["trade_penalty"])
    c.drawdown_penalty = float(cfg["drawdown_penalty"])
    c.passive_bonus = float(cfg["passive_bonus"])
    c.loss_multiplier = float(cfg["loss_multiplier"])
    return c, cfg


def dec_to_int_exp(d):
    """Dec struct (or numpy structured row) -> (sign, coefficient:int, exp)."""
    w = d.w if hasattr(d, "w") else d["w"]
    coeff = int(w[0]) | (int(w[1]) << 32) | (int(w[2]) << 64)
    sign = int(d.sign if hasattr(d, "sign") else d["sign"])
    exp = int(d.exp if hasattr(d, "exp") else d["exp"])
    return sign, coeff, exp


def dec_to_decimal(d):
    """Exact `decimal.Decimal` with the same sign/coefficient/exponent triple."""
    import decimal
    sign, coeff, exp = dec_to_int_exp(d)
    return decimal.Decimal((sign, tuple(int(ch) for ch in str(coeff)), exp))


def dec_to_str(d):
    """`str(Decimal)` of the triple - what info["NAV"] carries (info_helper.py:54)."""
    return str(dec_to_decimal(d))


def decimal_to_dec(x):
    """decimal.Decimal -> Dec struct (coefficient must be < 2^96)."""
    sign, digits, exp = x.as_tuple()
    coeff = int("".join(map(str, digits)) or "0")
    assert coeff < (1 << 96)
    d = Dec()
    d.w[0] = coeff & 0xFFFFFFFF
    d.w[1] = (coeff >> 32) & 0xFFFFFFFF
    d.w[2] = (coeff >> 64) & 0xFFFFFFFF
    d.exp = exp
    d.sign = sign
    return d
