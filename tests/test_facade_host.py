"""CPU: the dict facades' host-side marshalling (no GPU, no library call): the column-wise batch encoder writes exactly what the per-agent statement writes and hands
everything unusual to it; the snapshot decoder builds the reference's dicts from a block laid out like `packed`."""
import json

import numpy as np
import pytest
import torch

from gym_continuousdoubleauction_amd import _capi as K
from gym_continuousdoubleauction_amd.env import _ActionStage, _DictSurface, _Snap
from gym_continuousdoubleauction_amd.parallel import slab_layout


class _FakeVec:
    """what _init_surface reads of a CDAVecEnv"""

    def __init__(self, n, a, n_hist=4):
        self.config = {"num_of_agents": a, "max_step": 100, "n_hist": n_hist, "tick_size": 1, "is_render": False, "init_cash": 1000000}
        self.slab_layout = slab_layout(n, n_hist * 42, a)
        off, self.info_layout = self.slab_layout["bytes"], {}
        tdt = {K.C.c_int32: torch.int32, K.C.c_double: torch.float64, K.C.c_uint8: torch.uint8}
        for name, ct, per_agent, dims in K.INFO_FIELDS:
            shape = ((n, a) if per_agent else (n,)) + tuple(dims)
            shape = shape + (16,) if ct is K.Dec else shape
            dt = torch.uint8 if ct is K.Dec else tdt[ct]
            nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
            self.info_layout[name] = (off, dt, shape, nbytes)
            off += (nbytes + 15) // 16 * 16
        self.bytes = off


def surface(n, a):
    s = _DictSurface.__new__(_DictSurface)
    vec = _FakeVec(n, a)
    s._init_surface(vec)
    s._info_lay = vec.info_layout
    return s, vec


class _Stage(_ActionStage):
    def __init__(self, n, a):                      # the pinned block needs a GPU runtime: plain host memory here
        words = n * a
        self.host = torch.zeros(5 * 4 * words + (words + 15) // 16 * 16, dtype=torch.uint8)
        hn = self._hn = self.host.numpy()
        self.np = {k: hn[j * 4 * words:(j + 1) * 4 * words].view(dt).reshape(n, a) for j, (k, dt) in enumerate(self._FIELDS)}
        self.np["present"] = hn[20 * words:21 * words].reshape(n, a)
        self.flat = {k: v.reshape(-1) for k, v in self.np.items()}


def rand_dict(rng, agents, plain=False):
    f = (lambda x: float(x[0])) if plain else (lambda x: x)
    return {a: {"category": np.int64(rng.integers(0, 9)), "size_mean": f(rng.uniform(-1, 1, 1).astype(np.float32)), "size_sigma": f(rng.uniform(0, 1, 1).astype(np.float32)),
                "price": np.int64(rng.integers(0, 10)), "price_offset": np.int64(rng.integers(0, 3))} for a in agents}


def slow(s, dicts, n, a):
    st = _Stage(n, a)
    st.clear()
    for i, d in enumerate(dicts):
        s._encode(d, *(st.np[k][i] for k in ("category", "size_mean", "size_sigma", "price", "price_offset", "present")))
    return st.host.numpy().copy()


@pytest.mark.parametrize("n,a", [(1, 4), (7, 4), (5, 8)])
def test_batch_encoder_equals_the_per_agent_statement(n, a):
    s, _ = surface(n, a)
    rng = np.random.default_rng(n * 10 + a)
    st = _Stage(n, a)
    for variant in range(8):
        dicts = [rand_dict(rng, s.agents, plain=(variant == 6)) for _ in range(n)]
        if variant == 1:                                     # a key is missing -> defaults (price 0, price_offset 1)
            for d in dicts:
                del d["agent_1"]["price"], d["agent_2"]["price_offset"]
        if variant == 2:                                     # a subset of the agents acts
            del dicts[0]["agent_1"]
        if variant == 3:                                     # the dict's own order (reversed) travels in `present`
            dicts[-1] = dict(reversed(list(dicts[-1].items())))
        if variant == 4:                                     # a non-canonical key addresses the same agent
            d = dicts[0]
            dicts[0] = {("agent_02" if k == "agent_2" else k): v for k, v in d.items()}
        if variant == 5:                                     # float64 arrays, python ints
            for d in dicts:
                for v in d.values():
                    v["size_mean"] = v["size_mean"].astype(np.float64)
                    v["category"] = int(v["category"])
        if variant == 7:                                     # scalars of shape ()
            for d in dicts:
                for v in d.values():
                    v["size_sigma"] = np.float32(v["size_sigma"][0])
        st.host.fill_(0xAB)                                   # whatever the block held before must not leak into the step
        s._encode_all(dicts, st)
        assert np.array_equal(st.host.numpy()[:21 * n * a], slow(s, dicts, n, a)[:21 * n * a]), variant      # (behind the arrays: padding nobody reads)


def test_batch_encoder_raises_what_the_per_agent_statement_raises():
    s, _ = surface(3, 4)
    rng = np.random.default_rng(1)
    st = _Stage(3, 4)
    bad = [rand_dict(rng, s.agents) for _ in range(3)]
    bad[1]["agent_2"]["category"] = 9
    with pytest.raises(KeyError):
        s._encode_all(bad, st)
    bad = [rand_dict(rng, s.agents) for _ in range(3)]
    bad[2]["agent_0"]["size_sigma"] = np.array([-0.5], np.float32)
    with pytest.raises(ValueError, match="scale < 0"):
        s._encode_all(bad, st)
    bad = [rand_dict(rng, s.agents) for _ in range(3)]
    bad[0]["agent_9"] = bad[0]["agent_0"]
    with pytest.raises(KeyError):
        s._encode_all(bad, st)
    ok = [rand_dict(rng, s.agents) for _ in range(3)]
    ok[0]["agent_3"]["size_sigma"] = np.array([np.nan], np.float32)      # numpy's normal() takes a NaN scale: so do both encoders
    s._encode_all(ok, st)
    assert np.array_equal(st.host.numpy()[:21 * 12], slow(s, ok, 3, 4)[:21 * 12])


def test_snapshot_decode_builds_the_five_dicts_lob_actions_passes_and_bankrupts():
    s, vec = surface(1, 4)
    h = np.zeros(vec.bytes, np.uint8)
    lay = vec.slab_layout
    h[lay["obs"]:lay["obs"] + 168 * 4].view(np.float32)[:] = np.arange(168, dtype=np.float32)
    h[lay["reward"]:lay["reward"] + 32].view(np.float64)[:] = [0.5, -1.25, 0.0, 3.0]
    h[lay["truncated"]] = 1
    snap = _Snap(h, vec.info_layout)
    navs = ["1000000", "-12.5", "0", "999999.123456789"]
    import decimal
    for k, t in enumerate(navs):
        d = K.decimal_to_dec(decimal.Decimal(t))
        row = snap["nav"][0, k].view(K.DEC_DTYPE)
        row["w"][0] = list(d.w); row["exp"] = d.exp; row["sign"] = d.sign
    snap["is_pass_action"][0] = [0, 1, 0, 1]
    snap["lob_actions"][0] = [[0, 1, 5, 101], [-1, 0, 0, 0], [1, 0, 7, 0], [-1, 0, 0, 0]]
    snap["best_bid"][0] = np.nan
    snap["num_trades"][0] = [3, 0, 1, 2]
    rng = np.random.default_rng(0)
    actions = rand_dict(rng, s.agents)
    (obs, rew, term, trunc, infos), lob, passes, bankrupt = s._decode_snap(actions, h)
    assert obs["agent_0"] is obs["agent_3"] and np.array_equal(obs["agent_0"], np.arange(168, dtype=np.float32))
    assert rew == {"agent_0": 0.5, "agent_1": -1.25, "agent_2": 0.0, "agent_3": 3.0}
    assert term == {**dict.fromkeys(s.agents, False), "__all__": False} and trunc == {**dict.fromkeys(s.agents, False), "__all__": True}
    assert lob == [{"ID": "agent_0", "side": "bid", "type": "limit", "size": 5, "price": 101.0}, {"ID": "agent_2", "side": "ask", "type": "market", "size": 7, "price": 0.0}]
    assert passes == {"agent_1", "agent_3"} and bankrupt == {"agent_1", "agent_2"}          # NAV <= 0
    assert [infos[a]["NAV"] for a in s.agents] == navs and infos["agent_0"]["best_bid"] is None and infos["agent_2"]["num_trades"] == 1
    assert infos["agent_1"]["reward"] == -1.25 and json.loads(json.dumps(infos["agent_3"]))["model_action"]["category"] == int(actions["agent_3"]["category"])
    # the dict's own order decides the order of LOB_actions; a non-canonical key keeps its model_action
    rev = {("agent_02" if k == "agent_2" else k): v for k, v in reversed(list(actions.items()))}
    (_, _, _, _, infos2), lob2, _, _ = s._decode_snap(rev, h)
    assert [o["ID"] for o in lob2] == ["agent_2", "agent_0"] and infos2["agent_2"]["model_action"]["category"] == int(actions["agent_2"]["category"])
    # ... and it equals the eager decoder on the same block
    st = {k: snap[k] for k in vec.info_layout}
    o2 = h[lay["obs"]:lay["obs"] + 672].view(np.float32).reshape(1, 168)
    r2 = h[lay["reward"]:lay["reward"] + 32].view(np.float64).reshape(1, 4)
    (eo, er, et, etr, ei), elob, epass, ebank = s._decode(0, actions, o2, r2, h[lay["terminated"]:lay["terminated"] + 1], h[lay["truncated"]:lay["truncated"] + 1], st)
    assert er == rew and et == term and etr == trunc and elob == lob and epass == passes and ebank == bankrupt
    assert json.dumps(ei, sort_keys=True) == json.dumps(infos, sort_keys=True)


def test_fast_decimal_string_is_the_exact_triple():
    import decimal
    from gym_continuousdoubleauction_amd.env import _dec_str
    rng = np.random.default_rng(3)
    cases = ["0", "-0", "0.000", "-0.00", "1E+3", "1000000", "-12.5", "999999.123456789", "79228162514264337593543950335", "-7.9228162514264337593543950335E-5", "1E-28"]
    for _ in range(300):
        digits = int(rng.integers(1, 29))
        cases.append(("-" if rng.integers(2) else "") + str(int(rng.integers(1, 10))) + "".join(str(int(x)) for x in rng.integers(0, 10, digits - 1)) + f"E{int(rng.integers(-30, 5))}")
    arr = np.zeros(len(cases), K.DEC_DTYPE)
    for j, t in enumerate(cases):
        d = K.decimal_to_dec(decimal.Decimal(t))
        arr[j]["w"] = list(d.w); arr[j]["exp"] = d.exp; arr[j]["sign"] = d.sign
    for t, row, raw in zip(cases, arr.view(np.uint32).reshape(-1, 4).tolist(), arr):
        want = K.dec_to_decimal(raw)
        assert _dec_str(row) == str(want) and decimal.Decimal(_dec_str(row)).as_tuple() == want.as_tuple() == decimal.Decimal(t).as_tuple(), t
