"""Streams that really run concurrently.

HIP multiplexes its streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default); a stream is bound to a
queue at its first use and two streams that share a queue execute their kernels one after the other.  Measured on MI355X
(tools/stream_probe.py): of the pairs drawn from PyTorch's stream pool roughly one in four shares a queue, and a groups=2
env on such a pair runs at HALF speed (74 us per step instead of 37).  `concurrent_streams` therefore tests candidates
with two spin kernels and keeps a set whose members overlap pairwise; the result is cached per device, so every env of
the process steps its market groups on the same, known-good streams (two groups > 1 envs of one process therefore share
streams and serialise each other: give the second one its own via `CDAVecEnv(..., group_streams=[...])`).  When the pool
does not hold enough streams that verifiably overlap - more chains than hardware queues, or another process keeping the
GPU busy during the 5-ms test - the set is padded with unverified ones and a warning says so (expect up to half speed).
"""
import warnings
import time

import torch

_CACHE = {}


def _spin_pair_seconds(a, b, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    if b is not None:
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def concurrent_streams(device, n, candidates=12):
    """`n` streams of `device` that overlap pairwise (measured).  Falls back to the best effort (fewer verified streams,
    padded with unverified ones) if the pool does not hold `n` - the env then still works, only slower."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    have = _CACHE.setdefault(key, [])
    if len(have) >= n:
        return have[:n]
    with torch.cuda.device(device):
        pool = [torch.cuda.Stream(device=device) for _ in range(max(candidates, 2 * n))]
        for s in pool[:2]:                                   # first use binds the queue; also warms the spin kernel
            _spin_pair_seconds(s, None, 1000)
        probe = min(_spin_pair_seconds(pool[0], None, 100_000) for _ in range(2))
        cycles = int(min(20_000_000, max(100_000, 100_000 * 4e-4 / max(probe, 1e-6))))     # a spin of ~0.4 ms: far above launch noise
        one = min(_spin_pair_seconds(pool[0], None, cycles) for _ in range(2))
        chosen = list(have) or [pool[0]]
        for s in pool:
            if len(chosen) >= n:
                break
            if any(s.cuda_stream == c.cuda_stream for c in chosen):
                continue
            if all(min(_spin_pair_seconds(c, s, cycles) for _ in range(2)) < 1.5 * one for c in chosen):
                chosen.append(s)
        if len(chosen) < n:
            warnings.warn(f"only {len(chosen)} of the {n} requested streams were verified to run concurrently on {device} (GPU_MAX_HW_QUEUES defaults to 4; "
                          "a busy GPU also defeats the test): padding with unverified streams - market groups that share a hardware queue run one "
                          "after the other", RuntimeWarning, stacklevel=2)
        for s in pool:                                       # not enough verified streams: pad
            if len(chosen) >= n:
                break
            if not any(s.cuda_stream == c.cuda_stream for c in chosen):
                chosen.append(s)
    _CACHE[key] = chosen
    return chosen[:n]
