"""Instructions per market-step by phase and by action category, from PMC counters (run on the GPU box under rocprofv3):

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d DIR -o p --output-format csv -- \
        python tools/inst_count_probe.py <skip mask> <category 0..8 | 9 = uniform random>

Steps 4096 x 4 for 300 random steps, then 100 steps with the given debug skip mask (a -DCDA_DEBUG_SKIP build via CDA_HIP_LIB;
0 with the product build) and every agent playing `category`.  tools/inst_count_summary.py reads the last 100 k_step rows."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv, _lib  # noqa: E402

mask, cat_v = int(sys.argv[1]), int(sys.argv[2])
L = _lib.lib()
env = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}, 4096, with_info=False)
env.reset(seed=1000)
acts = env.random_actions_device(0, 400, action_seed=2024)
for t in range(300):
    env.step(*[x[t] for x in acts])
torch.cuda.synchronize()
if mask:
    L.cda_debug_set_skip.argtypes = [C.c_int]
    L.cda_debug_set_skip(mask)
cats = acts[0] if cat_v == 9 else torch.full_like(acts[0], cat_v)
torch.cuda.synchronize()
for t in range(300, 400):
    env.step(cats[t], acts[1][t], acts[2][t], acts[3][t], acts[4][t])
torch.cuda.synchronize()
