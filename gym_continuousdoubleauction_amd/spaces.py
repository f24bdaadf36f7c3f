"""Observation/action spaces of the env (action_helper.py:126-143, continuousDoubleAuction_env.py:109-119).

Uses `gymnasium.spaces` when it is importable so RLlib sees the real classes; otherwise a small
stand-in with the same attributes (`shape`, `dtype`, `n`, `low`, `high`, `spaces`, `sample`, `seed`,
`contains`) so the env is usable in images without gymnasium (this build image has none)."""
import numpy as np

try:  # (gymnasium is absent from the build image; tests/test_host_logic.py runs this branch under the stand-in of tests/golden/shim)
    from gymnasium import spaces as _gs
    Box, Discrete, Dict = _gs.Box, _gs.Discrete, _gs.Dict
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self):
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return seed

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            super().__init__()
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            lo, hi = self.low.astype(np.float64), self.high.astype(np.float64)
            if np.all(np.isfinite(lo)) and np.all(np.isfinite(hi)):
                return self._rng.uniform(lo, hi).astype(self.dtype)
            return self._rng.normal(size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    class Discrete(_Space):
        def __init__(self, n, start=0):
            super().__init__()
            self.n, self.start, self.shape, self.dtype = int(n), int(start), (), np.dtype(np.int64)

        def sample(self):
            return np.int64(self.start + self._rng.integers(self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

    class Dict(_Space):
        def __init__(self, spaces=None, **kw):
            super().__init__()
            self.spaces = dict(spaces or {})
            self.spaces.update(kw)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def seed(self, seed=None):
            ss = np.random.SeedSequence(seed)
            for child, sub in zip(self.spaces.values(), ss.spawn(len(self.spaces))):
                child._rng = np.random.default_rng(sub)
            return seed

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x):
            return all(k in x and s.contains(x[k]) for k, s in self.spaces.items())


def observation_space(n_hist):
    return Box(low=-np.inf, high=np.inf, shape=(n_hist * 42,), dtype=np.float32)


def action_space():
    """Dict{category: Discrete(9), size_mean: Box[-1,1](1,), size_sigma: Box[0,1](1,), price: Discrete(10),
    price_offset: Discrete(3)} - config/tunable_constants.json:16-21."""
    return Dict({
        "category": Discrete(9),
        "size_mean": Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32),
        "size_sigma": Box(low=0.0, high=1.0, shape=(1,), dtype=np.float32),
        "price": Discrete(10),
        "price_offset": Discrete(3),
    })
