"""Which stream pairs really run concurrently?  (run on the GPU box)  Steps a groups=2 env on explicitly chosen streams."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402

N, A, K = 4096, 4, 400
cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}
dev = torch.device("cuda:0")
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
pool = [torch.cuda.Stream(device=dev) for _ in range(12)]
print("stream handles:", [hex(s.cuda_stream) for s in pool])
env = CDAVecEnv(cfg, N, with_info=False, groups=2)
env.reset(seed=1000)
acts = env.random_actions_device(0, 64, action_seed=2024)


def run(pair):
    import ctypes as C
    env.group_streams = list(pair)
    env._stream_arr = (C.c_void_p * 2)(*[s.cuda_stream for s in pair])
    env.join(); torch.cuda.synchronize()
    for t in range(32):
        i = t % 64
        env.step(acts[0][i], acts[1][i], acts[2][i], acts[3][i], acts[4][i])
    env.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        i = t % 64
        env.step(acts[0][i], acts[1][i], acts[2][i], acts[3][i], acts[4][i])
    env.join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


if "--pick" in sys.argv:            # the package's own selection: verified-concurrent streams, cached per device
    from gym_continuousdoubleauction_amd.streams import concurrent_streams
    t0 = time.perf_counter()
    for s in pool[:7]:
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)   # other users of the stream pool came first
    ch = concurrent_streams(dev, 4)
    print(f"picked {[hex(s.cuda_stream) for s in ch]} in {time.perf_counter() - t0:.3f} s")
    for i in range(4):
        for j in range(i + 1, 4):
            print(f"picked ({i},{j}): {run((ch[i], ch[j])):6.1f} us/step", flush=True)
    sys.exit(0)
if "--prio" in sys.argv:            # does a (normal, high-priority) pair ever share a hardware queue?
    hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(6)]
    for i in range(6):
        for j in range(6):
            print(f"normal {i} + high {j}: {run((pool[i], hi[j])):6.1f} us/step", flush=True)
    for i in range(5):
        print(f"high {i} + high {i + 1}: {run((hi[i], hi[i + 1])):6.1f} us/step", flush=True)
    sys.exit(0)
if "--ramp" in sys.argv:            # the same 432 steps from the same reset, over and over: is there a clock ramp?
    for rep in range(14):
        env.reset(seed=1000)
        print(f"rep {rep}: {run((pool[0], pool[1])):6.1f} us/step", flush=True)
    sys.exit(0)
for i in range(0, 8):
    for j in range(i + 1, 9):
        print(f"streams ({i},{j}): {run((pool[i], pool[j])):6.1f} us/step", flush=True)
