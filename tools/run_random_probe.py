import sys, time, torch
sys.path.insert(0, '/root/repo')
from gym_continuousdoubleauction_amd import CDAVecEnv
for N, A in ((4096, 4), (16384, 4), (2048, 8)):
    env = CDAVecEnv({"num_of_agents": A, "init_cash": 1000000, "max_step": 1 << 20, "is_render": False}, n_markets=N, with_info=False)
    env.reset(seed=1)
    for steps in (64, 512):
        env.run_random(steps, action_seed=7); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.run_random(steps, action_seed=8); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        print(f"run_random {N} x {A}, {steps} steps in one launch: {ms:.2f} ms = {ms / steps * 1e3:.1f} us per step = {N * A * steps / ms / 1e3:.1f} M agent-steps/s", flush=True)
    env.close()
