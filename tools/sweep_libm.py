"""Build container (any x86-64 host with glibc): exhaustive / large-sample checks behind csrc/cda_libm.hpp.

1. the restated exp and log1p (the device source compiled for the host) against this machine's libm;
2. float32(log1p(M - 1)) == float32(numpy.log(M)) == float32(libm log(M)) for EVERY half-integer mid M = k/2, k <= 2^25
   (prices are below 2^24 ticks), which is what lets the observation's log(M) share the log1p code path with no fallback.
Result of the round-2 run (numpy 2.2.6, glibc 2.35, Xeon with AVX-512): 0 mismatches in all three.
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from gym_continuousdoubleauction_amd.vec_env import selftest_libm
rng = np.random.default_rng(1)
# 1. restated exp / log1p (host compile of the device source) vs this machine's libm
x = np.concatenate([-7.0 * rng.random(10_000_000), -0.01 * rng.random(4_000_000), 1024 * rng.random(6_000_000) - 512, np.array([0.0, -0.0, -1e-300, 1e-20, -1e-17])])   # domain |x| < 512
t = time.time(); a = selftest_libm(1, x, device=None); b = O.libm(1, x); print("exp mismatches", int((a.view(np.uint64) != b.view(np.uint64)).sum()), "of", len(x), round(time.time() - t, 1), "s")
x = np.concatenate([-rng.random(5_000_000), rng.random(2_000_000) * 1e6, rng.random(1_000_000) * 1e-3])
a = selftest_libm(0, x, device=None); b = O.libm(0, x); print("log1p mismatches", int((a.view(np.uint64) != b.view(np.uint64)).sum()), "of", len(x))
# 2. float32(log1p(M - 1)) vs float32(numpy.log(M)) and vs float32(libm log(M)) for EVERY half-integer mid below 2^24
bad_np, bad_lm = [], []
for lo in range(1, (1 << 25) + 1, 1 << 22):
    k = np.arange(lo, min(lo + (1 << 22), (1 << 25) + 1), dtype=np.float64)
    M = k / 2
    alt = selftest_libm(0, M - 1.0, device=None).astype(np.float32)
    ref = np.log(M).astype(np.float32)
    lm = O.libm(2, M).astype(np.float32)
    bad_np += list(k[alt.view(np.uint32) != ref.view(np.uint32)])
    bad_lm += list(k[alt.view(np.uint32) != lm.view(np.uint32)])
print("k <= 2^25: f32(log1p(M-1)) != f32(np.log(M)) at", len(bad_np), bad_np[:10], "; != f32(libm log) at", len(bad_lm), bad_lm[:10])
