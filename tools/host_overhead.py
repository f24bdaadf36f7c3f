"""Host-side (CPU) cost of the calls bench.py makes per step, measured as enqueue time of short bursts
(the GPU runs behind; nothing here waits for it).  Run on the GPU box:  python tools/host_overhead.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT))
from gym_continuousdoubleauction_amd import CDAVecEnv  # noqa: E402


def burst(fn, n=150, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        best = min(best, (time.perf_counter() - t0) / n)
        torch.cuda.synchronize()
    return best * 1e6


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    N, A = 4096, 4
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 100000, "is_render": False}
    env = CDAVecEnv(cfg, n_markets=N, with_info=False, out_buffers=2)
    env.reset(seed=1000)
    acts = env.random_actions_device(0, 64, action_seed=1)
    g = [torch.empty(env.slab_layout["bytes"], dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(400)]
    res = {}
    res["env.step"] = burst(lambda i: env.step(acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64]))
    if hasattr(env, "step_fast"):
        res["env.step_fast"] = burst(lambda i: env.step_fast(acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64]))
    res["index 5 action tensors"] = burst(lambda i: (acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64]))
    res["event.record"] = burst(lambda i: ev[i].record())
    works = []
    res["all_gather async"] = burst(lambda i: works.append(dist.all_gather_into_tensor(g[i & 1], env.out_slab, async_op=True)))
    res["work.wait"] = burst(lambda i: works[i].wait())
    res["all_gather sync"] = burst(lambda i: dist.all_gather_into_tensor(g[i & 1], env.out_slab))
    res["copy_ 2.9MB"] = burst(lambda i: g[i & 1].copy_(env.out_slab))
    for k, v in res.items():
        print(f"{k:28s} {v:8.1f} us/call")

    # ---- steady-state loop time (GPU and CPU together), us per step
    def loop(body, n=400):
        for i in range(20):
            body(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            body(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def st(i):
        env.step(acts[0][i % 64], acts[1][i % 64], acts[2][i % 64], acts[3][i % 64], acts[4][i % 64])

    side = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(4)]
    pend = [None]

    def v_events(i):
        ev[i % 400].record(); st(i); ev[(i + 1) % 400].record()

    def v_copy(i):
        st(i); g[i & 1].copy_(env.out_slab)

    def v_ag_sync(i):
        st(i); dist.all_gather_into_tensor(g[i & 1], env.out_slab)

    def v_ag_async(i):
        st(i)
        w = dist.all_gather_into_tensor(g[i & 1], env.out_slab, async_op=True)
        if pend[0] is not None:
            pend[0].wait()
        pend[0] = w

    def v_side_copy(i):
        st(i)
        e = evs[i & 1]
        e.record()
        with torch.cuda.stream(side):
            side.wait_event(e)
            g[i & 1].copy_(env.out_slab)
            evs[2 + (i & 1)].record()
        torch.cuda.current_stream().wait_event(evs[2 + ((i + 1) & 1)])

    # alternating streams: stream X runs step(t) then its gather (a synchronous collective stays on the current stream),
    # stream Y runs step(t+1) after ONE event edge on step(t); the gather of t overlaps step t+1 with one edge per step
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ev_step = [torch.cuda.Event() for _ in range(2)]
    state = {"first": True}

    def v_alt(i):
        cur = sa if (i & 1) == 0 else sb
        with torch.cuda.stream(cur):
            if not state["first"]:
                cur.wait_event(ev_step[(i + 1) & 1])          # market state: step i needs step i-1
            state["first"] = False
            st(i)
            ev_step[i & 1].record(cur)
            dist.all_gather_into_tensor(g[i & 1], env.out_slab)

    def v_alt_copy(i):
        cur = sa if (i & 1) == 0 else sb
        with torch.cuda.stream(cur):
            if not state["first"]:
                cur.wait_event(ev_step[(i + 1) & 1])
            state["first"] = False
            st(i)
            ev_step[i & 1].record(cur)
            g[i & 1].copy_(env.out_slab)

    for name, fn in (("step only", st), ("alternating streams + gather", v_alt), ("alternating streams + copy", v_alt_copy), ("step + 2 timing events", v_events), ("step + copy same stream", v_copy),
                     ("step + all_gather sync", v_ag_sync), ("step + all_gather async", v_ag_async),
                     ("step + copy side stream", v_side_copy)):
        print(f"loop: {name:28s} {loop(fn):8.1f} us/step")
    env.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
