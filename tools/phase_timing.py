#!/usr/bin/env python3
"""Debug: per-phase cycle breakdown of k_step (needs a -DCDA_PHASE_TIMING build of the library).
Run on the GPU box: python tools/phase_timing.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

so = os.path.join(ROOT, "gpurun_out", "libcda_hip_timing.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
COUNTERS = "--counters" in sys.argv      # atomics in every out-of-line decimal routine: call counts, but the cycle stamps are then meaningless
import __graft_entry__ as G   # the product's own flags + the timing macro
subprocess.check_call(["hipcc"] + G.HIPCC_FLAGS + ["-DCDA_PHASE_TIMING"] + (["-DCDA_DEC_COUNTERS"] if COUNTERS else []) + ["-o", so, os.path.join(ROOT, "gym_continuousdoubleauction_amd", "csrc", "cda_hip.hip"), os.path.join(ROOT, "gym_continuousdoubleauction_amd", "csrc", "cda_ppo.hip"),
                       os.path.join(ROOT, "gym_continuousdoubleauction_amd", "csrc", "cda_mlp.hip")])      # (the whole library: _lib binds every symbol of both headers)
from gym_continuousdoubleauction_amd import _lib
_lib.LIB_PATH = so
from gym_continuousdoubleauction_amd import CDAVecEnv

N, A = 4096, 4
env = CDAVecEnv({"num_of_agents": A, "init_cash": 1000000, "max_step": 100000, "is_render": False}, n_markets=N, with_info=False)
env.reset(seed=np.arange(1000, 1000 + N, dtype=np.uint64))
buf = torch.zeros((N, 40), dtype=torch.int64, device="cuda:0")
L = _lib.lib()
L.cda_debug_set_phase_buffer.argtypes = [C.c_void_p]
L.cda_debug_set_phase_buffer(C.c_void_p(buf.data_ptr()))
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
names = ["load", "snapshot_pre", "decode+rng", "shuffle", "orders", "mtm", "snapshot_post+obs", "reward/info", "store"]
acc = np.zeros(9); span = 0.0
tot_all = []; worst = None; sub = np.zeros(24); sub_worst = None; sub_slow = np.zeros(24); ph_slow = np.zeros(9)
T, W = 300, 200
calls = (C.c_ulonglong * 8)()
if COUNTERS:
    L.cda_debug_dec_calls.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for t in range(W + T):
    if t == W and COUNTERS:
        torch.cuda.synchronize(); L.cda_debug_dec_calls(calls, 1)
    cat = torch.randint(0, 9, (N, A), generator=g, device="cuda:0", dtype=torch.int32)
    price = torch.randint(0, 10, (N, A), generator=g, device="cuda:0", dtype=torch.int32)
    off = torch.randint(0, 3, (N, A), generator=g, device="cuda:0", dtype=torch.int32)
    mean = torch.rand((N, A), generator=g, device="cuda:0") * 2 - 1
    sigma = torch.rand((N, A), generator=g, device="cuda:0")
    env.step(cat, mean, sigma, price, off)
    if t >= W:
        b = buf.cpu().numpy().astype(np.float64)
        d = b[:, 1:10] - b[:, 0:9]
        acc += d.mean(axis=0)
        span += (b[:, 9].max() - b[:, 0].min())
        sub += b[:, 10:34].mean(axis=0)
        tw = d.sum(axis=1)
        tot_all.append(tw)
        i = int(tw.argmax())
        sub_slow += b[i, 10:34]; ph_slow += d[i]
        if worst is None or tw[i] > worst[0]:
            worst = (tw[i], d[i].copy(), t, i); sub_worst = b[i, 10:34].copy()
acc /= T
tot = acc.sum()
print(f"mean cycles per wave per step: {tot:.0f}; kernel span (first start -> last end) {span / T:.0f} cycles")
for n, v in zip(names, acc):
    print(f"  {n:20s} {v:10.0f} cycles  {100 * v / tot:5.1f} %")

tw = np.concatenate(tot_all)
print("per-wave total cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % (tw.mean(), *np.percentile(tw, [50, 90, 99, 99.9]), tw.max()))
per_step_max = np.array([x.max() for x in tot_all])
print("max over the 4096 waves of one step: mean %.0f  (min %.0f, max %.0f)" % (per_step_max.mean(), per_step_max.min(), per_step_max.max()))
print("slowest wave seen: %.0f cycles at step %d market %d; phases:" % (worst[0], worst[2], worst[3]), dict(zip(names, worst[1].astype(int).tolist())))

subn = ["approval", "find_own", "match+settle", "insert/remove(after match)", "cancel/escrow/other", "fills",
        "book_remove/in-place", "release transfer", "escrow transfer", "x9", "n_modify", "n_escrow", "lane-0 fills in mode 0", "x13",
        "fill:prep(mode,tv)", "fill:stage1 mul", "fill:stage2 select", "fill:stage2 add", "fill:stage3 (modes 1, 2)", "fill:tail(sync,ballots)", "fills with lane 0 involved", "lane-0 fills in mode 3/4", "fill:stage3 (covered)", "fill:stage3 (neutral)"]
print("orders phase breakdown, mean per wave per step:", {n: round(v / T, 1) for n, v in zip(subn, sub)})
print("orders phase breakdown, slowest wave:", dict(zip(subn, sub_worst.astype(int).tolist())))
print("the slowest wave of each step, averaged over the steps - phases:", dict(zip(names, (ph_slow / T).astype(int).tolist())))
print("  its orders phase:", {n: round(v / T, 1) for n, v in zip(subn, sub_slow)})

if not COUNTERS:
    sys.exit(0)
torch.cuda.synchronize(); L.cda_debug_dec_calls(calls, 0)
cn = ["d_fix_mid", "d_fix_wide", "d_round_mid", "d_add_wide", "d_add_mid", "d_div_general", "d_div_u32", "d_to_double_slow"]
print("out-of-line decimal calls per market-step (lanes counted individually):", {n: round(calls[i] / (T * N), 2) for i, n in enumerate(cn)})
