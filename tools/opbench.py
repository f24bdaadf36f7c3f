#!/usr/bin/env python3
"""Debug: cycles per decimal operation on representative ledger operands (single wave, dependent chain)."""
import ctypes as C
import os
import sys
from decimal import Decimal as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import _capi as K, _lib

def _tools_lib():
    """tools/libcda_tools.so: the probes are a library of their own, outside the product (built by __graft_entry__.build())"""
    import ctypes
    import os
    import torch  # noqa: F401  (its HIP runtime must be in the process first, see _lib.py)
    return ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcda_tools.so"))


L = _tools_lib()
L.cda_debug_opbench.argtypes = [C.c_int, C.c_int, C.POINTER(K.Dec), C.POINTER(K.Dec), C.c_void_p]
L.cda_debug_opbench.restype = C.c_longlong
ITERS = 2000
cases = [
    ("add  cash(28d,e-21) +- tv(6d,e-1)", 0, D("999525.000000000000000000001"), D("12345.0")),
    ("add  hold(6d,e-1) +- v(6d,e-1)    ", 0, D("52345.0"), D("12345.0")),
    ("add  big(28d,e-21) +- big(28d,e-21)", 0, D("999525.000000000000000000001"), D("123456.000000000000000000007")),
    ("add  raw(28d,e-23) +- profit(26d,e-22)", 0, D("99952.50000000000000000000001"), D("1234.5600000000000000000007")),
    ("mul  vwap(28d,e-26) * 1234 (rounds)", 1, D("56.99999999999999999999999999"), D(1234)),
    ("mul  price(3d,e-1) * 1234          ", 1, D("57.0"), D(1234)),
    ("div  num(28d,e-23) / 1234 (inexact)", 2, D("70337.99999999999999999999999"), D(1234)),
    ("div  num(7d,e-1) / 1234 (exact)    ", 2, D("70338.0"), D(1234)),
    ("cmp  cash(28d,e-21) vs val(6d,e-1) ", 3, D("999525.000000000000000000001"), D("12345.0")),
    ("f64  3.0000000000000000000000 (strip)", 4, D("3.0000000000000000000000"), D(1)),
    ("f64  56.99999999999999999999999999 (exact path)", 4, D("56.99999999999999999999999999"), D(1)),
    ("f64  12.5 (fast)", 4, D("12.5"), D(1)),
]
for name, op, a, b in cases:
    da, db = K.decimal_to_dec(a), K.decimal_to_dec(b)
    cyc = L.cda_debug_opbench(op, ITERS, C.byref(da), C.byref(db), None)
    print(f"{name:52s} {cyc / ITERS:9.1f} cycles/op")
