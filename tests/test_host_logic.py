"""CPU: host-side logic of the product (config mapping, Decimal triples, spaces, sharding helpers)."""
from decimal import Decimal

import numpy as np
import pytest
import torch

from gym_continuousdoubleauction_amd import _capi as K
from gym_continuousdoubleauction_amd import parallel as P
from gym_continuousdoubleauction_amd import spaces as S


def test_make_config_defaults_and_overrides():
    c, cfg = K.make_config(None)
    assert c.num_agents == 5 and c.max_step == 64 and c.init_cash == 1000000 and cfg["is_render"] is True
    c, _ = K.make_config({"num_of_agents": 8, "init_cash": 5000.0, "max_step": 4096, "loss_multiplier": 2.0})
    assert (c.num_agents, c.init_cash, c.max_step, c.loss_multiplier) == (8, 5000, 4096, 2.0)
    with pytest.raises(KeyError):
        K.make_config({"num_agents": 4})            # the env-side spelling is num_of_agents
    for bad in (0.5, 0, -3, 65537):                         # integer ticks 1 .. CDA_TICK_MAX only (include/cda.h)
        with pytest.raises(ValueError):
            K.make_config({"tick_size": bad})
    assert K.make_config({"tick_size": 5})[0].tick_size == 5 and K.make_config({"tick_size": 250.0})[0].tick_size == 250
    with pytest.raises(ValueError):
        K.make_config({"init_cash": 10.5})


@pytest.mark.parametrize("text", ["1000000", "999525.0", "978412.0000000000000000000002", "-12.50", "0E-27", "0.0",
                                  "1E+3", "0.000001234", "-0"])
def test_decimal_triple_roundtrip_and_str(text):
    d = Decimal(text)
    s = K.decimal_to_dec(d)
    back = K.dec_to_decimal(s)
    assert back.as_tuple() == d.as_tuple()
    assert K.dec_to_str(s) == str(d)


def test_spaces_match_the_reference_contract():
    obs = S.observation_space(4)
    assert obs.shape == (168,) and np.dtype(obs.dtype) == np.float32
    act = S.action_space()
    assert set(act.spaces.keys()) == {"category", "size_mean", "size_sigma", "price", "price_offset"}
    assert act["category"].n == 9 and act["price"].n == 10 and act["price_offset"].n == 3
    assert act["size_mean"].shape == (1,) and float(act["size_mean"].low[0]) == -1.0 and float(act["size_sigma"].low[0]) == 0.0
    act.seed(3)
    s = act.sample()
    assert act.contains(s) and s["size_mean"].dtype == np.float32


def test_shard_range_and_seeds():
    assert P.shard_range(0, 8, 16384) == (0, 2048) and P.shard_range(7, 8, 16384) == (14336, 2048)
    # a size the world does not divide: equal blocks, the last rank takes the remainder; the shards tile the batch
    assert [P.shard_range(r, 3, 10) for r in range(3)] == [(0, 3), (3, 3), (6, 4)] and P.shard_pad(3, 10) == 4
    for world, n in ((8, 16384), (3, 10), (7, 100), (5, 5)):
        parts = [P.shard_range(r, world, n) for r in range(world)]
        assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1)) and sum(c for _, c in parts) == n
    with pytest.raises(ValueError):
        P.shard_range(3, 3, 10)
    with pytest.raises(ValueError):
        P.shard_range(0, 4, 3)
    s = P.global_seeds(1000, 2048, 4)
    assert s.tolist() == [3048, 3049, 3050, 3051]


def test_pack_unpack_is_bit_preserving():
    g = torch.Generator().manual_seed(0)
    obs = torch.randn((5, 168), generator=g)
    rew = torch.randn((5, 4), generator=g, dtype=torch.float64) * 1e-9
    term = torch.tensor([0, 1, 0, 0, 1], dtype=torch.bool)
    trunc = torch.tensor([1, 0, 0, 1, 0], dtype=torch.bool)
    p = P.pack_outputs(obs, rew, term, trunc)
    assert p.shape == (5, 168 + 8 + 2)
    o2, r2, t2, u2 = P.unpack_outputs(p, 168, 4)
    assert torch.equal(o2.view(torch.int32), obs.view(torch.int32))
    assert torch.equal(r2.view(torch.int64), rew.view(torch.int64))
    assert torch.equal(t2, term) and torch.equal(u2, trunc)


def test_random_agent_sampler_matches_its_specification():
    """include/cda_random_agents.h through the library's host entry point vs an independent numpy statement of it."""
    import ctypes as C
    from gym_continuousdoubleauction_amd import _lib
    L = _lib.lib()
    n, a, step, seed, base = 37, 5, 11, 0xDEADBEEFCAFE, 1000
    cat, price, off = (np.zeros((n, a), np.int32) for _ in range(3))
    mean, sigma = (np.zeros((n, a), np.float32) for _ in range(2))
    assert L.cda_random_actions_host(seed, base, step, n, a, cat.ctypes.data, mean.ctypes.data, sigma.ctypes.data, price.ctypes.data, off.ctypes.data) == 0
    M = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9e3779b97f4a7c15) & M
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
        return z ^ (z >> 31)
    for i in range(n):
        for k in range(a):
            h0 = mix((seed + (base + i) * 0xd1342543de82ef95) & M)
            w0 = mix((h0 + ((step << 32) | k)) & M)
            w1 = mix(w0)
            assert cat[i, k] == ((w0 & 0xffffffff) * 9) >> 32 and price[i, k] == ((w0 >> 32) * 10) >> 32
            assert off[i, k] == ((w1 & 0xffffffff) * 3) >> 32
            assert mean[i, k] == np.float32(((w1 >> 32) & 0xffffff) / 8388608.0 - 1.0) and sigma[i, k] == np.float32((w1 >> 40) / 16777216.0)
    assert L.cda_random_actions_host(seed, base, -1, n, a, cat.ctypes.data, mean.ctypes.data, sigma.ctypes.data, price.ctypes.data, off.ctypes.data) != 0


def test_lazy_info_dicts_are_real_dicts_once_touched():
    """CDAVecMultiAgentEnv.step hands out N x A info dicts per step that are BUILT ON FIRST USE (env._LazyInfo).  Whatever touches
    them first - json.dumps (C encoder), ==, iteration, .get, len, dict(), {**d}, pickle, deepcopy - must see exactly what the eager
    builder (env._info_dict, the one CDAEnv uses) produces."""
    import copy
    import ctypes
    import json
    import pickle

    import numpy as np
    from gym_continuousdoubleauction_amd import _capi as K
    from gym_continuousdoubleauction_amd.env import _LazyInfo, _info_dict
    n, a = 3, 4
    rng = np.random.default_rng(0)
    info = {}
    for name, ct, per_agent, dims in K.INFO_FIELDS:
        shape = ((n, a) if per_agent else (n,)) + tuple(dims)
        if ct is K.Dec:
            info[name] = np.zeros(shape + (16,), np.uint8)
            info[name][..., 0] = rng.integers(1, 200, shape)
        else:
            info[name] = (rng.random(shape) * 5).astype({4: np.int32, 8: np.float64, 1: np.uint8}[ctypes.sizeof(ct)])
    info["best_bid"][1] = np.nan                                   # None in the dict
    act = {"category": np.int64(3), "size_mean": np.array([0.5], np.float32)}
    for (i, k, ma) in ((1, 2, act), (0, 0, None), (2, 3, act)):
        want = _info_dict(info, i, k, 0.25, ma)
        mk = lambda: _LazyInfo(info, i, k, 0.25, ma)   # noqa: E731
        assert dict.__len__(mk()) == 1                              # nothing but the reward has been built
        assert json.dumps(mk(), sort_keys=True) == json.dumps(want, sort_keys=True)
        assert json.dumps({"agent_0": mk()}) == json.dumps({"agent_0": want})
        assert mk() == want and want == mk() and not (mk() != want)
        assert list(mk()) == list(want) and list(mk().items()) == list(want.items()) and len(mk()) == len(want)
        assert mk()["NAV"] == want["NAV"] and mk().get("best_bid") == want["best_bid"] and "spread" in mk() and mk().get("nope", 7) == 7
        assert dict(mk()) == want and {**mk()} == want and copy.deepcopy(mk()) == want and pickle.loads(pickle.dumps(mk())) == want
        d = mk()
        d["extra"] = 1                                              # a consumer may add to it
        assert {k2: v for k2, v in d.items() if k2 != "extra"} == want


def test_handback_watchdog_exits_instead_of_hanging():
    """a native collective call that never returns, or whose work the device never finishes, ends the process with exit code 3
    and a diagnostic inside the deadline (parallel._guarded) - it does not hang until somebody kills the job."""
    import os
    import subprocess
    import sys
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import time, sys; sys.path.insert(0, %r)\n"
            "from gym_continuousdoubleauction_amd.parallel import _guarded\n"
            "print(_guarded(lambda: 41 + 1, 'quick call', timeout=5), flush=True)\n"
            "%s\n") % (ROOT, "%s")
    t0 = time.time()
    hang = subprocess.run([sys.executable, "-c", code % "_guarded(lambda: time.sleep(60), 'a collective that never returns', timeout=1.0)"], capture_output=True, text=True, timeout=120)
    assert hang.returncode == 3 and "[cda watchdog]" in hang.stderr and "never returns" in hang.stderr and hang.stdout.strip() == "42"
    assert time.time() - t0 < 45
    code2 = code % ("class S:\n    def query(self):\n        return False\n_guarded(lambda: None, 'device never finishes', streams=[S()], timeout=1.0)")
    stuck = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert stuck.returncode == 3 and "did not finish the enqueued collectives" in stuck.stderr
    # an exception inside the guarded call comes back on the caller's thread
    code3 = code % "try:\n    _guarded(lambda: 1 / 0, 'raises', timeout=5)\nexcept ZeroDivisionError:\n    print('raised')"
    ok = subprocess.run([sys.executable, "-c", code3], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and ok.stdout.split() == ["42", "raised"]


_PRESENT = """
import sys
sys.path.insert(0, {shim!r}); sys.path.insert(0, {root!r})
import gymnasium
from gymnasium.envs.registration import registry
from ray.rllib.env.multi_agent_env import MultiAgentEnv
import gym_continuousdoubleauction_amd as pkg
from gym_continuousdoubleauction_amd import spaces as S
from gym_continuousdoubleauction_amd.env import CDAEnv, CDAVecMultiAgentEnv
assert pkg.GYMNASIUM_REGISTERED and registry[pkg.ENV_ID]["entry_point"] == "gym_continuousdoubleauction_amd.env:CDAEnv"
assert issubclass(CDAEnv, MultiAgentEnv) and issubclass(CDAEnv, gymnasium.Env)
assert S.HAVE_GYMNASIUM and S.Box is gymnasium.spaces.Box and S.Discrete is gymnasium.spaces.Discrete and S.Dict is gymnasium.spaces.Dict
act, obs = S.action_space(), S.observation_space(4)
assert isinstance(act, gymnasium.spaces.Dict) and isinstance(act["category"], gymnasium.spaces.Discrete) and isinstance(act["size_mean"], gymnasium.spaces.Box)
assert isinstance(obs, gymnasium.spaces.Box) and tuple(obs.shape) == (168,)
act.seed(3)
for _ in range(20):
    a = act.sample()
    assert act.contains(a) and 0 <= int(a["category"]) < 9 and 0 <= int(a["price"]) < 10 and 0 <= int(a["price_offset"]) < 3
{gpu_part}
print("present-branch ok")
"""

_PRESENT_GPU = """
env = gymnasium.make(pkg.ENV_ID, config={"num_of_agents": 4, "init_cash": 1000000, "max_step": 16, "is_render": False})
assert isinstance(env, CDAEnv) and isinstance(env, MultiAgentEnv)
assert all(isinstance(env.action_spaces[a], gymnasium.spaces.Dict) and isinstance(env.observation_spaces[a], gymnasium.spaces.Box) for a in env.agents)
assert env.get_action_space("agent_0") is env.action_spaces["agent_0"]
o, info = env.reset(seed=5)
assert set(o) == set(env.agents) and all(env.observation_spaces[a].contains(o[a]) for a in env.agents)
sp = env.action_spaces["agent_0"]; sp.seed(11)
for t in range(16):
    o, r, term, trunc, info = env.step({a: sp.sample() for a in env.agents})
    assert set(r) == set(env.agents) and "__all__" in term and "__all__" in trunc
assert trunc["__all__"] is True
"""


def _run_present(gpu_part=""):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _PRESENT.format(shim=os.path.join(root, "tests", "golden", "shim"), root=root, gpu_part=gpu_part)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "present-branch ok" in out.stdout, (out.stdout[-800:], out.stderr[-1500:])


def test_gymnasium_and_rllib_present_branch_imports_and_registers():
    """with gymnasium and ray.rllib importable (the stand-ins of tests/golden/shim put FIRST on sys.path, in a fresh interpreter) the
    package takes the branches it cannot take in this image otherwise: CDAEnv subclasses RLlib's MultiAgentEnv, the spaces ARE gymnasium's
    classes, and the env is registered under the reference's id (gym_continuousDoubleAuction/__init__.py:18-21)."""
    _run_present()


@pytest.mark.gpu
def test_gymnasium_present_branch_steps_on_the_gpu():
    _run_present(_PRESENT_GPU)


def test_guarded_call_runs_on_the_ranks_device_and_a_slow_setup_raises(monkeypatch):
    """The current HIP device belongs to the host THREAD: the watchdog's worker selects the rank's device before the native call (otherwise every rank of a multi-GPU
    node would initialise its RCCL communicators on GPU 0), and the communicators' set-up deadline raises (the caller falls back to torch.distributed) instead of
    ending the process."""
    import threading
    import time
    import torch
    from gym_continuousdoubleauction_amd import parallel
    seen = {}

    def fake_set_device(d):
        seen["device"], seen["thread"] = torch.device(d), threading.get_ident()
    monkeypatch.setattr(torch.cuda, "set_device", fake_set_device)
    out = parallel._guarded(lambda: threading.get_ident(), "probe", timeout=5, device=torch.device("cuda", 3))
    assert seen["device"] == torch.device("cuda", 3) and seen["thread"] == out != threading.get_ident()      # selected inside the worker thread, before the call
    seen.clear()
    parallel._guarded(lambda: None, "cpu", timeout=5, device="cpu")
    assert not seen
    t0 = time.monotonic()
    try:
        parallel._guarded(lambda: time.sleep(30), "slow communicator set-up", timeout=0.5, on_timeout="raise")
        raise AssertionError("expected TimeoutError")
    except TimeoutError as e:
        assert "slow communicator set-up" in str(e) and time.monotonic() - t0 < 5
    assert parallel.COMM_INIT_TIMEOUT_S >= parallel.HANDBACK_TIMEOUT_S


def test_a_taken_gymnasium_id_is_left_alone_and_the_registration_switches_the_wrappers_off():
    """real gymnasium.register only warns on a duplicate id and overrides it - the import order would then decide what gymnasium.make returns.  The
    package checks the registry first; and its own registration asks for no passive checker / order enforcer (a multi-agent dict API is not what they wrap)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path.insert(0, {os.path.join(root, "tests", "golden", "shim")!r}); sys.path.insert(0, {root!r})
from gymnasium.envs.registration import register, registry
import gym_continuousdoubleauction_amd as first
assert first.GYMNASIUM_REGISTERED and registry[first.ENV_ID]["kwargs"] == {{"disable_env_checker": True, "order_enforce": False}}
del sys.modules["gym_continuousdoubleauction_amd"]
registry[first.ENV_ID] = {{"id": first.ENV_ID, "entry_point": "gym_continuousDoubleAuction.envs:continuousDoubleAuctionEnv", "kwargs": {{}}}}
import gym_continuousdoubleauction_amd as second
assert not second.GYMNASIUM_REGISTERED and registry[second.ENV_ID]["entry_point"].startswith("gym_continuousDoubleAuction.")
print("ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.stdout[-500:], out.stderr[-1500:])
