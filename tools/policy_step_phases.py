#!/usr/bin/env python3
"""Where k_policy_step's time goes: cycle stamps of every market-wave (a -DCDA_PHASE_TIMING build of the library, CDA_HIP_LIB=tools/libcda_phase.so) over a
rollout of the PPO loop's shape: entry -> tables + observation tile staged -> forward pass done -> actions sampled and recorded -> the step's phases.

    CDA_HIP_LIB=$PWD/tools/libcda_phase.so python tools/policy_step_phases.py [--markets 4096] [--agents 4] [--chains 4] [--steps 32]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--markets", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--steps", type=int, default=32)
    a = ap.parse_args()
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    from gym_continuousdoubleauction_amd._lib import lib
    L = lib()
    L.cda_debug_set_phase_buffer.argtypes = [C.c_void_p]
    N, T = a.markets, a.steps
    env = CDAVecEnv({"num_of_agents": a.agents, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}, n_markets=N, with_info=False)
    env.reset(seed=1)
    pol = mlp.FusedPolicy("cuda:0", seed=0)
    roll = mlp.RolloutChains(env, pol, T, groups=a.chains, seed=3, use_graphs=False)
    for _ in range(3):
        roll.run()
    torch.cuda.synchronize()
    buf = torch.zeros((N, 40), dtype=torch.int64, device="cuda:0")
    L.cda_debug_set_phase_buffer(C.c_void_p(buf.data_ptr()))
    roll.run()                                   # the stamps that survive are the LAST step's of every market
    torch.cuda.synchronize()
    L.cda_debug_set_phase_buffer(None)
    s = buf.cpu().numpy().astype(np.int64)
    seg = [("entry -> tables + observation tile in LDS (first barrier)", 34, 35), ("forward pass (3 layers, 3 barriers)", 35, 36), ("sampling + the rollout's records", 36, 37),
           ("record arrives, header decoded -> decode + rng", 37, 2), ("shuffle + orders + mark to market (2 -> 6)", 2, 6), ("aggregate + frame (6 -> 7)", 6, 7),
           ("reward / outputs (7 -> 8)", 7, 8), ("write-back (8 -> 9)", 8, 9), ("whole kernel, per wave (34 -> 9)", 34, 9)]
    print(f"k_policy_step, {N} markets x {a.agents} agents, {a.chains} chains, last step of a {T}-step rollout; shader cycles per market-wave: mean / p50 / p90 / max")
    for name, i, j in seg:
        d = s[:, j] - s[:, i]
        print(f"  {name:66s} {d.mean():9.0f} {np.percentile(d, 50):9.0f} {np.percentile(d, 90):9.0f} {d.max():9.0f}")
    per_wg = (s[:, 36] - s[:, 35]).reshape(-1, 16)
    print(f"forward pass per workgroup (its sixteen waves agree to within {int((per_wg.max(1) - per_wg.min(1)).mean())} cycles)")
    env.close()


if __name__ == "__main__":
    main()
