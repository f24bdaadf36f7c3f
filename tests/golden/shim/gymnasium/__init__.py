"""Minimal stand-in for `gymnasium`, used ONLY in the build container to import the
Python reference (read-only at /root/reference) when cutting golden vectors.

It provides the documented behaviour of gymnasium 1.2.2 (the reference's pin,
requirements-lock.txt:31) that the hot path relies on:

* ``Env.np_random`` is a ``numpy.random.Generator``;
* ``Env.reset(seed=s)`` with ``s is not None`` re-creates it as
  ``Generator(PCG64(SeedSequence(s)))`` (``gymnasium.utils.seeding.np_random``);
  ``seed=None`` keeps the current stream.

This file is test infrastructure written for this repo; it is not part of the
product and never travels to the GPU box as a dependency of anything.
"""
import numpy as np

from . import spaces  # noqa: F401


class Env:
    _np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return None

    def close(self):
        pass


def make(id, **kwargs):                            # noqa: A002
    from .envs.registration import make as _make
    return _make(id, **kwargs)
