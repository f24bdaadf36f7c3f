"""What engine clock does the GPU run at while it steps the env?  (run on the GPU box)
A one-wave spin kernel compares the shader-cycle counter with the constant 100 MHz counter: idle, and in between env steps."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv, _lib  # noqa: E402

def _tools_lib():
    """tools/libcda_tools.so: the probes are a library of their own, outside the product (built by __graft_entry__.build())"""
    import ctypes
    import os
    import torch  # noqa: F401  (its HIP runtime must be in the process first, see _lib.py)
    return ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcda_tools.so"))


L = _tools_lib()
L.cda_debug_clock_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
out = torch.zeros(3, dtype=torch.int64, device="cuda:0")


def probe(tag):
    L.cda_debug_clock_probe(200000, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c, w = int(out[0]), int(out[1])
    print(f"{tag:40s} shader cycles {c:9d}  100MHz ticks {w:7d}  -> {c / max(w, 1) * 100:7.1f} MHz", flush=True)


probe("cold")
probe("second probe")
env = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}, 4096, with_info=False, groups=2)
env.reset(seed=1000)
acts = env.random_actions_device(0, 64, action_seed=2024)
for rep in range(4):
    for k in range(20):
        for t in range(64):
            env.step(*[x[t] for x in acts])
    env.join()
    probe(f"after {1280 * (rep + 1)} env steps")
