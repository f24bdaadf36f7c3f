"""Prices the phases of k_step by leaving them out (needs a -DCDA_DEBUG_SKIP build: CDA_HIP_LIB=build_tmp/dbgskip.so).
Every agent passes, so the order phase is empty and the numbers are the FIXED cost of a step (run on the GPU box)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv, _lib  # noqa: E402

N, A = 4096, 4
cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}
L = _lib.lib()
L.cda_debug_set_skip.argtypes = [C.c_int]
L.cda_debug_set_skip.restype = None
CASES = [(0, "full step (all agents pass)"), (1 | 2, "- step_market and mark-to-market"), (1, "- step_market, mark-to-market kept"), (4, "- observation"),
         (8, "- reward"), (16, "- record write-back"), (1 | 2 | 4 | 8, "load + tables + outputs + store only"),
         (1 | 2 | 4 | 8 | 16, "load + tables only (no write-back)"), (32, "requests + table staging, then exit"), (64, "empty kernel")]
for groups in (1, 2):
    for mask, name in CASES:
        env = CDAVecEnv(cfg, N, with_info=False, groups=groups)
        env.reset(seed=1000)
        acts = env.random_actions_device(0, 300, action_seed=2024)
        for t in range(300):
            env.step(*[x[t] for x in acts])
        env.join()
        zero = torch.zeros_like(acts[0][0])
        L.cda_debug_set_skip(mask)
        torch.cuda.synchronize()
        # one captured graph of 50 steps: the host's ~12 us per launch is out of the picture
        import time
        main_s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(main_s):
            g.capture_begin()
            for s_ in env.group_streams:
                s_.wait_stream(main_s)
            env._need_fork = False
            for t in range(50):
                env.step(zero, acts[1][0], acts[2][0], acts[3][0], acts[4][0])
            for s_ in env.group_streams:
                main_s.wait_stream(s_)
            g.capture_end()
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                g.replay()
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400 * 1e6
        L.cda_debug_set_skip(0)
        print(f"groups={groups} {name:48s} {dt:7.1f} us/step", flush=True)
        env.close()
