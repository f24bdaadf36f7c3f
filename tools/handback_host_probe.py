"""Host cost of the pieces of one hand-back step (GPU box, one rank): what bounds `bench.py --force-gather`."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd.parallel import ShardedVecEnv  # noqa: E402
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402

dev = "cuda:0"
cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 100000, "is_render": False}
sh = ShardedVecEnv(cfg, 4096, device=dev, groups=4, handback=True)
sh.reset(seed_base=1000)
acts = sh.env.random_actions_device(0, 64, action_seed=1)
torch.cuda.synchronize()


def timed(name, fn, n=400):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    th = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    print(f"{name:60s} host {th * 1e6:7.1f} us/step   wall {t * 1e6:7.1f} us/step")


env = sh.env
step = lambda i: env.step(acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64], pipelined=True)  # noqa: E731
timed("env.step pipelined (cda_step_groups, 4 chains)", step)
print("transport:", sh.transport)
timed("sh.step pipelined, native (ONE call: 4 x (k_step, gather, unpack))", lambda i: sh.step(acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64], pipelined=True))
sh.transport = "torch"
timed("sh.step pipelined, torch path (4 x (stream ctx, copy, unpack))", lambda i: sh.step(acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64], pipelined=True))
s0 = sh.group_streams[0]


def ctx(i):
    for g in range(4):
        with torch.cuda.stream(sh.group_streams[g]):
            pass


timed("4 x torch.cuda.stream context only", ctx)
first, cnt = sh.group_ranges[0]


def copy4(i):
    for g in range(4):
        f, c = sh.group_ranges[g]
        sh._gbuf[g][0].copy_(env.handback[f:f + c])


timed("4 x buf.copy_(records slice)", copy4)


def unpack4(i):
    for g in range(4):
        f, c = sh.group_ranges[g]
        sh._unpack(sh._gbuf[g], 1, c, sh.n_local, f, sh.num_agents, sh.n_hist, *sh.full)


timed("4 x cda_handback_unpack (ctypes)", unpack4)
