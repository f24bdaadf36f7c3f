#!/usr/bin/env python3
"""Dict-facade latency: env-steps/s of CDAEnv and CDAVecMultiAgentEnv(256) through the dict protocol, host-resident step I/O (default) against the
staged-copy path (CDA_FACADE_HOST_IO=0), and where a CDAEnv.step's microseconds go.   python tools/facade_probe.py  (on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 100000, "is_render": False}


def rand_dict(rng, agents):
    return {a: {"category": np.int64(rng.integers(0, 9)), "size_mean": rng.uniform(-1, 1, 1).astype(np.float32), "size_sigma": rng.uniform(0, 1, 1).astype(np.float32),
                "price": np.int64(rng.integers(0, 10)), "price_offset": np.int64(rng.integers(0, 3))} for a in agents}


def rate(mode):
    os.environ["CDA_FACADE_HOST_IO"] = mode
    from gym_continuousdoubleauction_amd import CDAEnv, CDAVecMultiAgentEnv
    rng = np.random.default_rng(0)
    env = CDAEnv(CFG)
    env.reset(seed=1)
    acts = [rand_dict(rng, env.agents) for _ in range(200)]
    for k in range(50):
        env.step(acts[k])
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        for k in range(200):
            env.step(acts[k])
        best = max(best, 200 / (time.perf_counter() - t0))
    # and a consumer that reads every info dict
    t0 = time.perf_counter()
    for k in range(200):
        out = env.step(acts[k])
        json.dumps(out[4]["agent_0"]); len(out[4]["agent_1"]); len(out[4]["agent_2"]); len(out[4]["agent_3"])
    touched = 200 / (time.perf_counter() - t0)
    phases = None
    if mode == "1":                                              # where the microseconds go
        st, vec = env._stage, env._vec
        T = dict(encode=0.0, launch=0.0, sync=0.0, snapshot=0.0, decode=0.0)
        for k in range(200):
            a = acts[k]
            t0 = time.perf_counter(); env._encode_all((a,), st)
            t1 = time.perf_counter(); vec.step_host_io(st.host_ptrs)
            t2 = time.perf_counter(); vec.sync_host_io()
            t3 = time.perf_counter(); h = env._hio.copy()
            t4 = time.perf_counter(); env._decode_snap(a, h)
            t5 = time.perf_counter()
            for n, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                T[n] += d
        phases = {n: round(v / 200 * 1e6, 2) for n, v in T.items()}
    env.close()
    n = 256
    me = CDAVecMultiAgentEnv(dict(CFG, max_step=1000), num_envs=n)
    me.reset(seed=1)
    batch = [[rand_dict(rng, me.agents) for _ in range(n)] for _ in range(8)]
    me.step(batch[0]); me.step(batch[1])
    bm = 0.0
    for _ in range(4):
        t0 = time.perf_counter()
        for k in range(8):
            me.step(batch[k])
        bm = max(bm, 8 * n / (time.perf_counter() - t0))
    vph = None
    if mode == "1":
        st, vec = me._stage, me._vec
        T = dict(encode=0.0, launch=0.0, sync=0.0, rest=0.0)
        for k in range(8):
            t0 = time.perf_counter(); me._encode_all(batch[k], st)
            t1 = time.perf_counter(); vec.step_host_io(st.host_ptrs)
            t2 = time.perf_counter(); vec.sync_host_io()
            t3 = time.perf_counter()
            for nme, d in zip(T, (t1 - t0, t2 - t1, t3 - t2)):
                T[nme] += d
        t0 = time.perf_counter()
        for k in range(8):
            me.step(batch[k])
        T["rest"] = (time.perf_counter() - t0) - T["encode"] - T["launch"] - T["sync"]
        vph = {nme: round(v / 8 * 1e6, 1) for nme, v in T.items()}
    me.close()
    return {"host_io": mode == "1", "CDAEnv_env_steps_per_s": round(best), "CDAEnv_every_info_dict_read": round(touched), "CDAEnv_us_per_step": round(1e6 / best, 1),
            "CDAVecMultiAgentEnv256_env_steps_per_s": round(bm), "CDAEnv_phases_us": phases, "vec256_phases_us_per_step": vph}


if __name__ == "__main__":
    import subprocess
    if len(sys.argv) > 1:
        print(json.dumps(rate(sys.argv[1])))
    else:
        for mode in ("0", "1", "0", "1"):
            print(subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True).stdout.strip().splitlines()[-1])
        print("reference (pure Python, SURVEY section 6): 2,260 env-steps/s; round 5: CDAEnv 8,850, CDAVecMultiAgentEnv(256) 200,000")
