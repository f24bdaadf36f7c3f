"""GPU: device ledger arithmetic vs CPython `decimal`, device RNG vs numpy (both live third-party
implementations of the arithmetic the reference delegates to - SURVEY §8c)."""
import random
import struct
from decimal import Decimal as D

import numpy as np
import pytest

import oracle_lib as O
from gym_continuousdoubleauction_amd import _capi as K

pytestmark = pytest.mark.gpu


from dec_cases import rnd_dec, make_pairs  # noqa: E402,F401


@pytest.mark.parametrize("op,name", [(0, "add"), (1, "sub"), (2, "mul"), (3, "div"), (4, "cmp"), (5, "float")])
def test_device_decimal_matches_cpython(op, name):
    from gym_continuousdoubleauction_amd.vec_env import selftest_dec
    rng = random.Random(100 + op)
    n = 20000
    A, B = make_pairs(rng, n, op)
    if op == 5:   # the domain of the device conversion: exponent in [-109, 0]
        A = A + [rnd_dec(rng, emin=-109, emax=-40) for _ in range(4000)] + [D("8.333333333333333333333333333E-28")]
        B = B + B[:4001]
    a, b = O.dec_array(A), O.dec_array(B)
    out = selftest_dec(op, a, b)
    ref = O.dec_op(op, a, b)            # the CPU oracle must agree as well
    bad = []
    for i in range(len(A)):
        if op == 0:
            exp = A[i] + B[i]
        elif op == 1:
            exp = A[i] - B[i]
        elif op == 2:
            exp = A[i] * B[i]
        elif op == 3:
            exp = A[i] / B[i]
        elif op == 4:
            c = (A[i] > B[i]) - (A[i] < B[i])
            if int(out[i]["w"][0]) - 1 != c:
                bad.append((A[i], B[i], c, int(out[i]["w"][0]) - 1))
            continue
        else:
            bits = struct.unpack("<Q", struct.pack("<d", float(A[i])))[0]
            got = int(out[i]["w"][0]) | (int(out[i]["w"][1]) << 32)
            if got != bits or int(out[i]["w"][2]) != 0:
                bad.append((A[i], hex(bits), hex(got)))
            continue
        got = K.dec_to_decimal(out[i])
        if got.as_tuple() != exp.as_tuple():
            bad.append((A[i], B[i], exp, got))
    assert not bad, bad[:5]
    if op < 4:
        assert np.array_equal(out.view(np.uint8), ref.view(np.uint8))


def test_device_rng_matches_numpy():
    from gym_continuousdoubleauction_amd.vec_env import selftest_rng
    for seed in [0, 1, 2, 3, 123, 977, 2 ** 32 + 5, 2 ** 63 + 12345, 2 ** 64 - 1]:
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        first, normals, perms, fs = selftest_rng(seed, 10, 100, 400, 8, 8)
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


def test_device_normal_slow_paths_match_numpy():
    """Long stream: exercises the ziggurat wedge and tail branches (log1p / exp on the device)."""
    from gym_continuousdoubleauction_amd.vec_env import selftest_rng
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(4242)))
    first, normals, perms, fs = selftest_rng(4242, 0, 0, 1, 200000, 0)
    g.integers(0, 1)
    z = g.standard_normal(200000)
    assert np.array_equal(z.view(np.uint64), normals[0].view(np.uint64))


def test_device_libm_restatements_equal_the_host_libm():
    """glibc's log1p and exp as the DEVICE evaluates them (csrc/cda_libm.hpp) against the host's libm, bit for bit: the two
    functions numpy's normal sampler calls outside its fast path, plus the observation's log(M) = log1p(M - 1) at float32."""
    from gym_continuousdoubleauction_amd.vec_env import selftest_libm
    rng = np.random.default_rng(8)
    x = np.concatenate([-7.0 * rng.random(1_500_000), -0.01 * rng.random(300_000), 1024 * rng.random(200_000) - 512, np.array([0.0, -0.0, -1e-300, -6.676])])
    assert np.array_equal(selftest_libm(1, x).view(np.uint64), O.libm(1, x).view(np.uint64))
    x = np.concatenate([-rng.random(1_000_000), rng.random(500_000) * 1e6, np.array([0.0, 1.0, 2.0 ** 24])])
    assert np.array_equal(selftest_libm(0, x).view(np.uint64), O.libm(0, x).view(np.uint64))
    M = rng.integers(1, (1 << 25) + 1, 1_000_000).astype(np.float64) / 2
    assert np.array_equal(selftest_libm(0, M - 1.0).astype(np.float32).view(np.uint32), np.log(M).astype(np.float32).view(np.uint32))
    # the size rows: sqrt of an integer level volume (< 2^30, include/cda.h config bound) - every volume up to 2^22 and a sample above
    v = np.concatenate([np.arange(1, (1 << 22) + 1, dtype=np.float64), rng.integers(1 << 22, 1 << 31, 3_000_000).astype(np.float64)])
    assert np.array_equal(selftest_libm(2, v).view(np.uint64), np.sqrt(v).view(np.uint64))


def test_device_float_conversion_near_rounding_boundaries():
    """float(Decimal) on the device has a certified double-double path and an exact integer path behind it.
    28-digit decimals that sit ON, next to and at graded distances from the midpoint of two adjacent doubles
    exercise the certificate (reject -> exact path) and the accepted side of it."""
    from gym_continuousdoubleauction_amd.vec_env import selftest_dec
    import decimal
    rng = random.Random(77)
    ctx = decimal.Context(prec=28, rounding=decimal.ROUND_HALF_EVEN)
    vals = []
    while len(vals) < 30000:
        mag = rng.choice([1e-9, 1e-3, 1.0, 57.0, 1e3, 999999.0, 1e6, 1.2345e7, 1e12])
        x = abs(rng.gauss(0, 1)) * mag + mag * 1e-3
        m = D(x) + D(np.spacing(x)) / 2                     # the exact midpoint above x (a long decimal)
        base = ctx.plus(m)                                  # its nearest 28-digit decimal: 1e-28 relative ~ 2^-93 of the boundary
        e = base.as_tuple().exponent
        if e > 0 or e < -44:
            continue
        step = D((0, (1,), e))
        for dlt in (0, 1, -1, 2, -3, 10 ** rng.randint(1, 12), -(10 ** rng.randint(1, 12))):
            v = base + dlt * step
            if len(v.as_tuple().digits) <= 28 and v > 0:
                vals.append(v if rng.random() < 0.5 else -v)
    vals += [D(2) ** 60, D(2) ** 70 + 1, D("9007199254740993"), D("9007199254740992.000000000001"), D("1.000000000000000055511151231")]
    a = O.dec_array(vals)
    out = selftest_dec(5, a, a)
    bad = []
    for i, v in enumerate(vals):
        bits = struct.unpack("<Q", struct.pack("<d", float(v)))[0]
        got = int(out[i]["w"][0]) | (int(out[i]["w"][1]) << 32)
        if got != bits or int(out[i]["w"][2]) != 0:
            bad.append((v, hex(bits), hex(got)))
    assert not bad, (len(bad), bad[:5])


def _order_value(rng):
    """'p.0' * n as the ledger forms it: coefficient 10 * p * n, exponent -1."""
    p, n = rng.randint(1, 5000), rng.randint(1, 20000)
    return D(p) * D(n) * D("1.0")


def test_transfer_leaf_matches_cpython():
    """op 6: field + order value through d_add_order_value (cash / cash_on_hold transfers), any field - positive, negative, zero
    with an exponent of its own, short or 28 digits - and a share of operands the leaf must hand back to the general addition."""
    from gym_continuousdoubleauction_amd.vec_env import selftest_dec
    rng = random.Random(606)
    A, B = [], []
    for i in range(30000):
        r = rng.random()
        if r < 0.35:                                   # a cash that went through a division: 28 digits around 1e6
            f = (D(rng.randint(1, 2 * 10 ** 6)) + D(rng.randint(0, 10 ** 21)).scaleb(-21)) * 1
        elif r < 0.55:                                 # a cash_on_hold: short, exponent -1
            f = D(rng.randint(0, 10 ** 7)).scaleb(-1)
        elif r < 0.65:                                 # zero with an exponent
            f = D((rng.randint(0, 1), (0,), rng.randint(-27, 0)))
        else:
            f = rnd_dec(rng)
        if rng.random() < 0.3:
            f = -f
        v = _order_value(rng) if rng.random() < 0.85 else rnd_dec(rng)
        if rng.random() < 0.5:
            v = -v
        if rng.random() < 0.03:
            v = -f                                      # exact cancellation (whatever the shape)
        A.append(f); B.append(v)
    a, b = O.dec_array(A), O.dec_array(B)
    out = selftest_dec(6, a, b)
    bad = [(A[i], B[i], A[i] + B[i], K.dec_to_decimal(out[i])) for i in range(len(A)) if K.dec_to_decimal(out[i]).as_tuple() != (A[i] + B[i]).as_tuple()]
    assert not bad, bad[:5]
    ref = O.dec_op(0, a, b)                          # and the CPU oracle's addition, field by field (`pad` carries the leaf's flag here)
    assert all(np.array_equal(out[k], ref[k]) for k in ("w", "exp", "sign"))
    assert 0.25 < float(out["pad"].mean()) < 0.99      # both the leaf and its fallback were exercised
