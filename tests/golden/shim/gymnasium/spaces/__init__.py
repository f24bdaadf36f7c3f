"""Space stand-ins: shape/bounds containers with the sampling law only."""
import numpy as np


class Space:
    def __init__(self):
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return seed


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        super().__init__()
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def sample(self):
        lo = np.broadcast_to(np.asarray(self.low, dtype=np.float64), self.shape)
        hi = np.broadcast_to(np.asarray(self.high, dtype=np.float64), self.shape)
        if np.all(np.isfinite(lo)) and np.all(np.isfinite(hi)):
            return self._rng.uniform(lo, hi, self.shape).astype(self.dtype)
        return self._rng.normal(size=self.shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__()
        self.n, self.start, self.shape, self.dtype = int(n), int(start), (), np.int64

    def sample(self):
        return np.int64(self.start + self._rng.integers(self.n))

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        super().__init__()
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, key):
        return self.spaces[key]

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


class Tuple(Space):
    def __init__(self, spaces):
        super().__init__()
        self.spaces = tuple(spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)
