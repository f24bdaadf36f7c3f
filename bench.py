#!/usr/bin/env python3
"""bench.py - agent-steps/sec of the vectorised CDA env step on N MI355X GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one env step() of every market of the batch (one launch of the k_step kernel per rank).
Workload (BASELINE.json configs[2], the configuration the >=1M agent-steps/s target is quoted on):
4096 independent markets x 4 synthetic random agents PER GPU (weak scaling: markets are sharded,
there is no collective on the simulation path); for N > 1 the per-rank obs/reward shards are
all-gathered over RCCL each step - the hand-back to a central learner that north_star names.
Actions are pre-generated on the device (uniform random-agent law, train/model/model_handler.py:38-53
of the reference) so that the timed region starts with every input resident in HBM.

Prints ONE JSON line on rank 0; see the fields `roofline` and `cpu_baseline` in DESIGN.md.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_CONST = 1444      # B(A) = 1444 + 324*A bytes per market-step
ALG_BYTES_PER_AGENT = 324
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=64)
    p.add_argument("--markets", type=int, default=4096, help="markets per GPU")
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--info", action="store_true", help="also emit the info tensors every step")
    p.add_argument("--no-gather", action="store_true", help="N>1: skip the obs/reward all-gather")
    p.add_argument("--no-overlap", action="store_true", help="N>1: wait for each step's all-gather before the next launch")
    p.add_argument("--force-gather", action="store_true",
                   help="run the N>1 code path (process group + all-gather) even with one rank; diagnostics")
    p.add_argument("--fused", type=int, default=0, metavar="T",
                   help="not the headline run: T steps per launch through cda_run_random (random agents sampled in the kernel, "
                        "market state resident in LDS across steps, no per-step barrier between markets)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work budget of the cpu_baseline sample")
    return p.parse_args()


def gen_actions(torch, n, a, steps, device, seed):
    """Uniform random-agent law, generated on the device, [steps, n, a] per field."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cat = torch.randint(0, 9, (steps, n, a), generator=g, device=device, dtype=torch.int32)
    price = torch.randint(0, 10, (steps, n, a), generator=g, device=device, dtype=torch.int32)
    off = torch.randint(0, 3, (steps, n, a), generator=g, device=device, dtype=torch.int32)
    mean = torch.rand((steps, n, a), generator=g, device=device, dtype=torch.float32) * 2.0 - 1.0
    sigma = torch.rand((steps, n, a), generator=g, device=device, dtype=torch.float32)
    return cat, mean, sigma, price, off


def cpu_baseline(markets, agents, budget_s, max_step):
    """The CPU oracle (a plain-C port of the reference path, bit-identical to it on the golden
    vectors) timed on the host cores: markets partitioned over one thread per core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = markets
    cfg = {"num_of_agents": agents, "init_cash": 1000000, "max_step": max_step, "is_render": False}
    env = O.OracleEnv(cfg, n_markets=n)
    env.reset(seeds=np.arange(1000, 1000 + n, dtype=np.uint64))
    rng = np.random.default_rng(2024)
    T = 32
    cat = rng.integers(0, 9, (T, n, agents)).astype(np.int32)
    mean = rng.uniform(-1, 1, (T, n, agents)).astype(np.float32)
    sigma = rng.uniform(0, 1, (T, n, agents)).astype(np.float32)
    price = rng.integers(0, 10, (T, n, agents)).astype(np.int32)
    off = rng.integers(0, 3, (T, n, agents)).astype(np.int32)
    lib = O.lib()
    cores = min(cores, n)
    bounds = [(i * n // cores, (i + 1) * n // cores) for i in range(cores)]
    bounds = [(lo, hi) for lo, hi in bounds if hi > lo]

    def run_steps(count):
        def work(lo, hi):      # ONE foreign call per thread; ctypes releases the GIL for its duration
            lib.oracle_run_range(env.h, lo, hi - lo, count, T, cat.ctypes.data, mean.ctypes.data, sigma.ctypes.data,
                                 price.ctypes.data, off.ctypes.data, env.obs.ctypes.data, env.reward.ctypes.data,
                                 env.term.ctypes.data, env.trunc.ctypes.data)
        th = [threading.Thread(target=work, args=b) for b in bounds]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return time.perf_counter() - t0

    dt = run_steps(8)                      # calibration
    per_step = dt / 8
    steps = int(max(16, min(4000, budget_s / max(per_step, 1e-6))))
    dt = run_steps(steps)
    value = n * agents * steps / dt
    env.close()
    return {"value": value, "unit": "agent-steps/s", "cores": len(bounds), "kind": "port",
            "sample": f"{n} markets x {agents} agents x {steps} steps ({dt:.1f} s wall), C oracle, "
                      f"{len(bounds)} threads, obs+reward outputs only"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N GPUs but started as a single process: relaunch as one rank per GPU (what the driver does itself)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # diagnostics only: CDA_BENCH_DEVICE pins every rank to one GPU and CDA_BENCH_BACKEND=gloo replaces RCCL, so that the
    # multi-rank code path can be exercised on a single-GPU box (tests/test_bench_contract.py)
    if os.environ.get("CDA_BENCH_DEVICE"):
        local_rank = int(os.environ["CDA_BENCH_DEVICE"])
    backend = os.environ.get("CDA_BENCH_BACKEND", "nccl")
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from gym_continuousdoubleauction_amd import CDAVecEnv

    N, A, K, W = args.markets, args.agents, args.steps, args.warmup
    max_step = max(4096, K + W + 1)                   # no truncation inside the run
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": max_step, "is_render": False}
    if args.fused and (args.steps % args.fused or args.warmup % args.fused):
        raise SystemExit("--fused T needs --steps and --warmup to be multiples of T")
    gather = use_dist and not args.no_gather and not args.fused
    # N > 1: the per-step outputs (obs | reward | flags: one contiguous slab written by k_step itself) are
    # all-gathered over xGMI with ONE collective per step and no packing pass.  The env rotates two slabs, so
    # the gather of step t runs on RCCL's stream underneath the kernel of step t+1.
    env = CDAVecEnv(cfg, n_markets=N, device=str(device), with_info=args.info, out_buffers=2 if gather else 1)
    first_market = rank * N                           # global market index -> seed, independent of the GPU count
    seeds = (1000 + first_market + torch.arange(N, dtype=torch.int64)).numpy().astype("uint64")
    env.reset(seed=seeds)
    # action stream: chunks of <= 256 steps keep the resident set small (20 B per agent-step)
    chunk = min(256, K + W)
    acts = gen_actions(torch, N, A, chunk, device, 2024 + rank)
    if gather:
        gathered = [torch.empty(world * env.slab_layout["bytes"], dtype=torch.uint8, device=device) for _ in range(2)]
    # N > 1, overlapped: two streams alternate.  Stream X runs step t and then its all-gather (a synchronous collective
    # stays on the caller's stream); stream Y runs step t+1 as soon as ONE event says step t is done (the market state
    # dependency), i.e. underneath gather t.  Step t+2 is back on X behind gather t, which is also what protects the
    # slab gather t reads.  One cross-stream edge per step (~12 us on this platform, tools/host_overhead.py) instead of
    # the two that an async_op collective on RCCL's own stream costs.
    overlap = gather and not args.no_overlap
    if overlap:
        streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
        step_done = [torch.cuda.Event(), torch.cuda.Event()]
    timed = {}                                                 # global step index -> (start, end) timing events

    def one_step(t):
        i = t % chunk
        if args.fused:
            if t % args.fused == 0:
                env.run_random(args.fused, action_seed=2024 + rank, market_index_base=first_market)
            return
        if not overlap:
            ev = timed.get(t)
            if ev:
                ev[0].record()
            env.step(acts[0][i], acts[1][i], acts[2][i], acts[3][i], acts[4][i])
            if ev:
                ev[1].record()
            if gather:
                dist.all_gather_into_tensor(gathered[t & 1], env.out_slab)
            return
        cur = streams[t & 1]
        with torch.cuda.stream(cur):
            if t > 0:
                cur.wait_event(step_done[(t - 1) & 1])
            ev = timed.get(t)
            if ev:
                ev[0].record(cur)
            env.step(acts[0][i], acts[1][i], acts[2][i], acts[3][i], acts[4][i])
            if ev:
                ev[1].record(cur)
            step_done[t & 1].record(cur)
            dist.all_gather_into_tensor(gathered[t & 1], env.out_slab)

    torch.cuda.synchronize()                                   # reset and the action stream are complete before any side stream starts
    try:
        for t in range(W):
            one_step(t)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001 - the overlapped schedule could not be exercised on a multi-GPU node before the driver's run
        if not overlap:
            raise
        print(f"[bench] overlapped all-gather failed in warm-up ({e}); falling back to the serial schedule", file=sys.stderr)
        overlap = False
        torch.cuda.synchronize()
        for t in range(W):
            one_step(t)
        torch.cuda.synchronize()
    # Which all-gather schedule is faster depends on what the transfer costs on THIS node's links against ~12 us of stream
    # dependency per step; nothing could measure that before the driver's multi-GPU run, so both are timed for a few
    # (untimed) steps and every rank adopts the one whose slowest rank is faster.
    T0 = W                                                     # global index of the first timed step
    schedule_note = None
    if gather and overlap and not args.fused:
        cal = 32
        took = []
        for mode in (False, True):
            overlap = mode
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for t in range(cal):
                one_step(T0 + t)
            torch.cuda.synchronize()
            took.append(time.perf_counter() - c0)
            T0 += cal
        tk = torch.tensor(took, dtype=torch.float64, device=device)
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        took = [float(x) for x in tk.tolist()]
        overlap = took[1] <= took[0]
        schedule_note = f"calibrated over {cal} steps each: serial {took[0] / cal * 1e6:.1f} us/step, overlapped {took[1] / cal * 1e6:.1f} us/step"
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the stream the kernel is launched on (torch's current stream == the stream handed
    # to cda_step).  N == 1: one pair brackets the K back-to-back launches (nothing else is enqueued in
    # between), so kernel_ms = span / K.  N > 1: the all-gather and the stream dependencies sit between launches, so
    # single launches get their own pair - every EV_STRIDE-th one only, a timing event pair costs ~7 us of
    # stream time (tools/host_overhead.py).
    per_launch = gather
    EV_STRIDE = 16
    if per_launch:
        timed.update({T0 + t: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for t in range(0, K, EV_STRIDE)})
    else:
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if not per_launch:
        ev_a.record()
    for t in range(K):
        one_step(T0 + t)
    if not per_launch:
        ev_b.record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = (sum(a.elapsed_time(b) for a, b in timed.values()) / len(timed)) if per_launch else ev_a.elapsed_time(ev_b) / K
    flags = env.flags()
    n_flagged = int((flags != 0).sum().item())

    if rank == 0:
        total_agent_steps = float(world) * N * A * K
        value = total_agent_steps / elapsed
        alg_bytes = (ALG_BYTES_CONST + ALG_BYTES_PER_AGENT * A) * N            # per launch (one rank)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, calibrated on a known byte count in
        # this library's access pattern by tools/profile_gpu.sh); a committed measurement of THIS workload, not live
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as fh:
                pmc = json.load(fh)
            if N == 4096 and A == 4 and not args.info and not args.fused:
                traffic, traffic_src = pmc["hbm_bytes_per_launch"], "profiles/pmc_traffic_latest.json (rocprofv3 --pmc, calibrated)"
        except Exception:  # noqa: BLE001
            pass
        out = {
            "metric": "agent-steps/sec (whole node), 4 agents x N parallel markets",
            "value": value, "unit": "agent-steps/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+dec28+f64", "data": "synthetic",
            "config": {"workload": f"{N} markets x {A} random agents per GPU, book capacity {256} resting orders per market "
                                   f"(BASELINE configs[2]); global {world * N} markets"
                                   + (f"; FUSED: {args.fused} steps per launch (cda_run_random)" if args.fused else ""),
                       "markets_per_gpu": N, "agents": A, "info_outputs": bool(args.info), "gather_schedule": schedule_note,
                       "collective": ("all_gather(obs|reward|flags slab), " + ("overlapped with the next step on alternating streams" if overlap else "serial")) if gather else "none",
                       "flagged_markets": n_flagged},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_run_random (per step)" if args.fused else "k_step", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(N, A, args.cpu_seconds, max_step)
            except Exception as e:  # noqa: BLE001 - the baseline is a reported extra, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    env.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
