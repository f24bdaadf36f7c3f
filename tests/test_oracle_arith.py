"""CPU: the oracle's own arithmetic against the live third-party implementations the reference delegates to
(SURVEY §8c): CPython `decimal` (prec 28, ROUND_HALF_EVEN) and numpy's Generator(PCG64(SeedSequence(seed)))."""
import random
import struct
from decimal import Decimal as D

import numpy as np
import pytest

import oracle_lib as O
from dec_cases import make_pairs, rnd_dec
from gym_continuousdoubleauction_amd import _capi as K


@pytest.mark.parametrize("op,name", [(0, "add"), (1, "sub"), (2, "mul"), (3, "div"), (4, "cmp"), (5, "float")])
def test_oracle_decimal_matches_cpython(op, name):
    rng = random.Random(100 + op)
    n = 20000
    A, B = make_pairs(rng, n, op)
    if op == 5:   # exponent domain of the conversion: [-109, 0]
        A = A + [rnd_dec(rng, emin=-109, emax=-40) for _ in range(4000)] + [D("8.333333333333333333333333333E-28")]
        B = B + B[:4001]
    out = O.dec_op(op, O.dec_array(A), O.dec_array(B))
    bad = []
    for i in range(len(A)):
        if op == 4:
            c = (A[i] > B[i]) - (A[i] < B[i])
            if int(out[i]["w"][0]) - 1 != c:
                bad.append((A[i], B[i], c))
            continue
        if op == 5:
            bits = struct.unpack("<Q", struct.pack("<d", float(A[i])))[0]
            got = int(out[i]["w"][0]) | (int(out[i]["w"][1]) << 32)
            if got != bits:
                bad.append((A[i], hex(bits), hex(got)))
            continue
        exp = A[i] + B[i] if op == 0 else A[i] - B[i] if op == 1 else A[i] * B[i] if op == 2 else A[i] / B[i]
        got = K.dec_to_decimal(out[i])
        if got.as_tuple() != exp.as_tuple():
            bad.append((A[i], B[i], exp, got))
    assert not bad, (len(bad), bad[:5])


def test_oracle_rng_matches_numpy():
    for seed in [0, 1, 2, 3, 123, 977, 2 ** 32 + 5, 2 ** 63 + 12345, 2 ** 64 - 1]:
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        first, normals, perms, fs = O.rng_schedule(seed, 10, 100, 400, 8, 8)
        assert first == int(g.integers(10, 101))
        for s in range(400):
            z = np.array([g.standard_normal() for _ in range(8)])
            p = g.permutation(8)
            assert np.array_equal(z.view(np.uint64), normals[s].view(np.uint64)), (seed, s)
            assert np.array_equal(p, perms[s]), (seed, s)
        st = g.bit_generator.state
        assert (int(fs[0]) << 64 | int(fs[1])) == st["state"]["state"]
        assert (int(fs[2]) << 64 | int(fs[3])) == st["state"]["inc"]
        assert int(fs[4]) == st["has_uint32"]


def test_oracle_normal_slow_paths_match_numpy():
    """Long stream: the ziggurat wedge and tail branches (libm log1p / exp)."""
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(4242)))
    first, normals, perms, fs = O.rng_schedule(4242, 0, 0, 1, 200000, 0)
    g.integers(0, 1)
    z = g.standard_normal(200000)
    assert np.array_equal(z.view(np.uint64), normals[0].view(np.uint64))


def test_restated_libm_functions_equal_the_host_libm():
    """csrc/cda_libm.hpp restates glibc's log1p (ziggurat tail) and exp (ziggurat wedge) for the device; the SAME source compiled
    for the host is compared here with the machine's own libm - what numpy calls - bit for bit (exp: the FMA build glibc
    selects on every current x86-64 CPU; domain |x| < 512, the wedge needs [-6.7, 0])."""
    from gym_continuousdoubleauction_amd.vec_env import selftest_libm
    rng = np.random.default_rng(5)
    x = np.concatenate([-7.0 * rng.random(3_000_000), -0.01 * rng.random(1_000_000), 1024 * rng.random(1_000_000) - 512,
                        np.array([0.0, -0.0, -1e-300, 1e-20, -1e-17, -6.676, -0.5])])
    assert np.array_equal(selftest_libm(1, x, device=None).view(np.uint64), O.libm(1, x).view(np.uint64))
    x = np.concatenate([-rng.random(2_000_000), rng.random(1_000_000) * 1e6, rng.random(500_000) * 1e-3, np.array([0.0, 1.0, 0.5, 2.0 ** 24])])
    assert np.array_equal(selftest_libm(0, x, device=None).view(np.uint64), O.libm(0, x).view(np.uint64))


def test_log_of_half_integer_mid_equals_log1p_form_at_float32():
    """The device computes the observation's log(M) as log1p(M - 1) (one code path shared with the spread feature, no log()).
    For the half-integer mids M = k/2 that is the same float32 as numpy.log(M) and as libm's log(M): every k <= 2^20 and a
    two-million sample of k <= 2^25 (all mids of prices below 2^24 ticks) here; tools/sweep_libm.py swept all 2^25 once."""
    from gym_continuousdoubleauction_amd.vec_env import selftest_libm
    rng = np.random.default_rng(6)
    k = np.concatenate([np.arange(1, (1 << 20) + 1, dtype=np.float64), rng.integers(1 << 20, (1 << 25) + 1, 2_000_000).astype(np.float64),
                        np.array([float(1 << 25), float((1 << 25) - 1)])])
    M = k / 2
    alt = selftest_libm(0, M - 1.0, device=None).astype(np.float32)
    assert np.array_equal(np.log(M).astype(np.float32).view(np.uint32), alt.view(np.uint32))
    assert np.array_equal(O.libm(2, M).astype(np.float32).view(np.uint32), alt.view(np.uint32))
