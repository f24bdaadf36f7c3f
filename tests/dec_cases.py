"""Random Decimal operands shaped like the ledger's values (shared by the CPU oracle tests and the GPU device tests)."""
from decimal import Decimal as D


def rnd_dec(rng, maxdig=28, emin=-30, emax=0, allow_zero=True):
    nd = rng.randint(1, maxdig)
    if allow_zero and rng.random() < 0.05:
        c = 0
    else:
        c = rng.randint(10 ** (nd - 1) if nd > 1 else 1, 10 ** nd - 1)
    r = rng.random()
    if r < 0.2:
        keep = max(1, nd // 2)
        c = int(str(c)[:keep] + "0" * (nd - keep))
    elif r < 0.3:
        c = int("9" * nd)
    elif r < 0.4 and nd > 2:
        c = int("1" + "0" * (nd - 2) + "5")
    return D((rng.randint(0, 1), tuple(map(int, str(c))), rng.randint(emin, emax)))


def make_pairs(rng, n, op):
    A = [rnd_dec(rng) for _ in range(n)]
    B = []
    for a in A:
        if op in (2, 3):
            c = rng.randint(1, 2 ** 32 - 1) if rng.random() < 0.5 else rng.randint(1, 5000)
            e = -1 if (op == 2 and rng.random() < 0.5) else 0
            B.append(D((0, tuple(map(int, str(c))), e)))
            continue
        r = rng.random()
        if r < 0.3:
            B.append(rnd_dec(rng))
        elif r < 0.6:
            e = a.as_tuple().exponent
            B.append(rnd_dec(rng, emin=e - 3, emax=min(0, e + 3)))
        elif r < 0.7:
            B.append(-a if op == 0 else a)
        else:
            B.append(a + D((rng.randint(0, 1), (rng.randint(1, 9),), a.adjusted() - rng.randint(20, 40))))
    return A, B
