#!/usr/bin/env python3
"""Launch the PMC calibration kernels: 10 reads and 10 writes of 1 GiB (past the 256 MiB Infinity Cache),
4 B per lane coalesced.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gym_continuousdoubleauction_amd import _lib

def _tools_lib():
    """tools/libcda_tools.so: the probes are a library of their own, outside the product (built by __graft_entry__.build())"""
    import ctypes
    import os
    import torch  # noqa: F401  (its HIP runtime must be in the process first, see _lib.py)
    return ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcda_tools.so"))


L = _tools_lib()
L.cda_debug_calib.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
n_bytes = 1 << 30
buf = torch.zeros(n_bytes // 4, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
for _ in range(10):
    L.cda_debug_calib(C.c_void_p(buf.data_ptr()), n_bytes, 0, None)
for _ in range(10):
    L.cda_debug_calib(C.c_void_p(buf.data_ptr()), n_bytes, 1, None)
torch.cuda.synchronize()
print("calib done", n_bytes)
