#!/usr/bin/env python3
"""bench.py - agent-steps/sec of the vectorised CDA env step on N MI355X GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one env step() of every market of the batch.  Workload (BASELINE.json configs[2], the configuration the
>=1M agent-steps/s target is quoted on; `--config c3`): 4096 independent markets x 4 synthetic random agents PER GPU
(weak scaling: markets are sharded, there is no collective on the simulation path); `--config c4` is the per-GPU share of
BASELINE configs[3] (2048 markets x 8 agents).  For N > 1 the per-rank obs/reward shards are all-gathered over RCCL each
step - the hand-back to a central learner that north_star names.

Inputs: the action of (step, market, agent) comes from the counter-based generator of
include/cda_random_agents.h keyed (2024, step, GLOBAL market index, agent) - the uniform random-agent law of the
reference (train/model/model_handler.py:38-53).  The whole stream is generated on the device before the timed region
(cda_random_actions), so every input is resident in HBM; the cpu_baseline leg replays the SAME stream.

On one GPU the batch is stepped as `--groups` (default 4; 2 for runs shorter than 200 steps) contiguous market groups, each a chain of k_step launches on
its own stream (cda_step_groups): markets never interact, so the batch-wide barrier of a single launch is not part of
the reference's semantics, and a group's slowest market then overlaps the other group's work.  `roofline.kernel_ms` comes from
HIP event pairs on the group streams themselves: on every chain, or (`--event-lanes one`, the default below 200 timed steps)
on the first chain only - a timing event pair costs each stream ~7 us, which a 0.7-ms run notices.

Prints ONE JSON line on rank 0; see the fields `roofline` and `cpu_baseline` in DESIGN.md.
"""
import argparse
import datetime
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
GPU_CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md: peak engine clock
N_SIMD = 1024               # 256 CUs x 4 SIMDs
VALU_CYCLES = 2.0           # MI355X_MICROARCH.md: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles
PG_TIMEOUT = datetime.timedelta(seconds=240)      # a collective that never completes must fail the run, not hang it (the default is 10 minutes)
ACTION_SEED = 2024          # bench_seed
SEED_BASE = 1000            # market i is seeded SeedSequence(1000 + i)
CONFIGS = {"c3": (4096, 4), "c4": (2048, 8)}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=64)
    p.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="c3: 4096 markets x 4 agents per GPU (BASELINE configs[2]); "
                   "c4: 2048 x 8 per GPU (the per-GPU share of BASELINE configs[3], 16384 x 8 over 8 GPUs)")
    p.add_argument("--markets", type=int, default=None, help="markets per GPU (overrides --config)")
    p.add_argument("--agents", type=int, default=None)
    p.add_argument("--groups", type=int, default=None, help="concurrent market groups per GPU (default 4, 2 below 200 timed steps; 1 with the all-gather)")
    p.add_argument("--event-lanes", choices=["all", "one"], default=None,
                   help="HIP event pairs on every group stream, or on the first one only (default: all, one below 200 timed steps)")
    p.add_argument("--info", action="store_true", help="headline run WITH the info tensors (a14); otherwise info-on is timed as a second leg")
    p.add_argument("--no-info-leg", action="store_true", help="skip the second (info tensors on) timed leg")
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(markets, agents, budget_s, max_step, first_market):
    """The CPU oracle (a plain-C port of the reference path, bit-identical to it on the golden vectors) timed on the
    host cores on the SAME action stream the GPU leg consumes (random agents keyed (2024, step, global market, agent),
    same market seeds): all cores (markets partitioned, one foreign call per thread) and one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = markets
    cfg = {"num_of_agents": agents, "init_cash": 1000000, "max_step": max_step, "is_render": False}
    lib = O.lib()
    seeds = np.arange(SEED_BASE + first_market, SEED_BASE + first_market + n, dtype=np.uint64)

    def timed(n_markets, n_threads, steps):
        env = O.OracleEnv(cfg, n_markets=n_markets)
        env.reset(seeds=seeds[:n_markets])
        bounds = [(i * n_markets // n_threads, (i + 1) * n_markets // n_threads) for i in range(n_threads)]
        bounds = [(lo, hi) for lo, hi in bounds if hi > lo]

        def work(lo, hi):      # ONE foreign call per thread; ctypes releases the GIL for its duration
            lib.oracle_run_random_range(env.h, lo, hi - lo, 0, steps, ACTION_SEED, first_market, env.obs.ctypes.data, env.reward.ctypes.data,
                                        env.term.ctypes.data, env.trunc.ctypes.data)
        th = [threading.Thread(target=work, args=b) for b in bounds]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        env.close()
        return dt, len(bounds)

    threads = min(cores, n)
    dt, _ = timed(n, threads, 8)                                   # calibration on a throw-away env
    steps = int(max(16, min(max_step - 1, 0.8 * budget_s / max(dt / 8, 1e-6))))
    dt, used = timed(n, threads, steps)                            # steps 0 .. steps-1 of the GPU leg's stream
    value = n * agents * steps / dt
    n1 = min(n, 64)
    dt1, _ = timed(n1, 1, 4)
    steps1 = int(max(8, min(max_step - 1, 0.2 * budget_s / max(dt1 / 4, 1e-6))))
    dt1, _ = timed(n1, 1, steps1)
    value1 = n1 * agents * steps1 / dt1
    return {"value": value, "unit": "agent-steps/s", "cores": used, "kind": "port", "value_1thread": value1, "cpu_model": cpu_model(),
            "sample": f"{n} markets x {agents} agents x steps 0..{steps - 1} of the GPU leg's action stream ({dt:.1f} s wall), C oracle, "
                      f"{used} threads, obs+reward outputs only; 1 thread: {n1} markets x {steps1} steps ({dt1:.1f} s)"}


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
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"), timeout=PG_TIMEOUT)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=PG_TIMEOUT)
    n_gpus = world
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from gym_continuousdoubleauction_amd import CDAVecEnv

    N, A = CONFIGS[args.config]
    N = args.markets if args.markets is not None else N
    A = args.agents if args.agents is not None else A
    K, W = args.steps, args.warmup
    if args.fused and (args.steps % args.fused or args.warmup % args.fused):
        raise SystemExit("--fused T needs --steps and --warmup to be multiples of T")
    gather = use_dist and not args.no_gather and not args.fused
    CAL = 32 if gather and not args.no_overlap else 0             # untimed steps of each all-gather schedule (calibration)
    total_steps = W + 2 * CAL + K
    max_step = max(4096, total_steps + 1)                          # no truncation inside the run
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": max_step, "is_render": False}
    # default number of group chains: 4 pays once the chains are long (433 M vs 417 M at 1000 steps), 2 when the whole timed
    # region is a few dozen steps and the staggered start / drain of four chains is a visible share of it (384 M vs 372 M at 20)
    event_lanes = args.event_lanes or ("all" if K >= 200 else "one")
    groups = args.groups if args.groups is not None else (1 if (gather or args.fused) else (4 if K >= 200 else 2))
    groups = max(1, min(groups, N))
    if gather and groups != 1:
        raise SystemExit("the all-gather schedule steps the shard as one launch: use --groups 1")
    first_market = rank * N                           # global market index -> seed and action key, independent of the GPU count
    seeds = (SEED_BASE + first_market + torch.arange(N, dtype=torch.int64)).numpy().astype("uint64")

    def make_env(with_info):
        # N > 1: the per-step outputs (obs | reward | flags: one contiguous slab written by k_step itself) are all-gathered
        # over xGMI with ONE collective per step and no packing pass; the env rotates two slabs, so the gather of step t
        # runs underneath the kernel of step t+1.
        e = CDAVecEnv(cfg, n_markets=N, device=str(device), with_info=with_info, out_buffers=2 if gather else 1, groups=groups)
        e.reset(seed=seeds)
        return e

    # Clock primer: an idle MI355X sits in a low-power state and needs tens of milliseconds of work before its engine clock
    # is up; the driver's default run times 20 steps (~1 ms) right after start-up, which would measure the ramp, not the
    # kernel.  A SCRATCH env (not the measured one) is stepped for ~PRIMER_MS first; the measured env then does exactly
    # W untimed + K timed steps from its own reset.
    primer_ms = float(os.environ.get("CDA_BENCH_PRIMER_MS", "40"))
    primer_steps = 0
    if primer_ms > 0 and not args.fused:
        scratch = CDAVecEnv(cfg, n_markets=N, device=str(device), with_info=False, groups=groups)
        scratch.reset(seed=seeds)
        pa = scratch.random_actions_device(0, 64, action_seed=ACTION_SEED + 1, market_index_base=first_market)
        t_end = time.perf_counter() + primer_ms * 1e-3
        while time.perf_counter() < t_end:
            for i in range(64):
                scratch.step(pa[0][i], pa[1][i], pa[2][i], pa[3][i], pa[4][i], pipelined=True)
            scratch.join()
            torch.cuda.synchronize()
            primer_steps += 64
        scratch.close()
        del pa
    env = make_env(args.info)
    # The action stream of the whole run, resident in HBM (20 B per agent-step: 350 MB for the default 1064 steps of
    # 4096 x 4); beyond MAX_RESIDENT steps the run cycles through the first MAX_RESIDENT (reported in `data`).
    MAX_RESIDENT = 4096
    period = min(total_steps, MAX_RESIDENT)
    acts = None if args.fused else env.random_actions_device(0, period, action_seed=ACTION_SEED, market_index_base=first_market)
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

    def one_step(e, t):
        i = t % period
        if args.fused:
            if t % args.fused == 0:
                e.run_random(args.fused, action_seed=ACTION_SEED, market_index_base=first_market)
            return
        if not overlap:
            ev = timed.get(t)
            if ev:
                ev[0].record()
            e.step(acts[0][i], acts[1][i], acts[2][i], acts[3][i], acts[4][i], pipelined=True)
            if ev:
                ev[1].record()
            if gather:
                dist.all_gather_into_tensor(gathered[t & 1], e.out_slab)
            return
        cur = streams[t & 1]
        with torch.cuda.stream(cur):
            if t > 0:
                cur.wait_event(step_done[(t - 1) & 1])
            ev = timed.get(t)
            if ev:
                ev[0].record(cur)
            e.step(acts[0][i], acts[1][i], acts[2][i], acts[3][i], acts[4][i], pipelined=True)
            if ev:
                ev[1].record(cur)
            step_done[t & 1].record(cur)
            dist.all_gather_into_tensor(gathered[t & 1], e.out_slab)

    def agree(flag):
        """N > 1: every rank must take the same branch (a rank-local decision would deadlock the next collective)."""
        if not use_dist:
            return flag
        v = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return bool(v.item() > 0)

    torch.cuda.synchronize()                                   # reset and the action stream are complete before any side stream starts
    failed = False
    try:
        for t in range(W):
            one_step(env, t)
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001 - the overlapped schedule could not be exercised on a multi-GPU node before the driver's run
        if not overlap:
            raise
        print(f"[bench] rank {rank}: overlapped all-gather failed in warm-up ({ex})", file=sys.stderr)
        failed = True
    if overlap and agree(failed):
        # a collective that raised leaves the process group unusable: rebuild it, then every rank takes the serial schedule
        print("[bench] falling back to the serial all-gather schedule on a fresh process group", file=sys.stderr)
        overlap, CAL = False, 0
        torch.cuda.synchronize()
        dist.destroy_process_group()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=PG_TIMEOUT)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=PG_TIMEOUT)
        env.close()
        env = make_env(args.info)
        torch.cuda.synchronize()
        for t in range(W):
            one_step(env, t)
        torch.cuda.synchronize()
    # Which all-gather schedule is faster depends on what the transfer costs on THIS node's links against ~12 us of stream
    # dependency per step; nothing could measure that before the driver's multi-GPU run, so both are timed for a few
    # (untimed) steps and every rank adopts the one whose slowest rank is faster.
    T0 = W                                                     # global index of the first timed step
    schedule_note, schedule_us = None, None
    if gather and overlap:
        took = []
        for mode in (False, True):
            overlap = mode
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for t in range(CAL):
                one_step(env, T0 + t)
            torch.cuda.synchronize()
            took.append(time.perf_counter() - c0)
            T0 += CAL
        tk = torch.tensor(took, dtype=torch.float64, device=device)
        dist.all_reduce(tk, op=dist.ReduceOp.MAX)
        took = [float(x) for x in tk.tolist()]
        overlap = took[1] <= took[0]
        schedule_us = {"serial_us_per_step": took[0] / CAL * 1e6, "overlapped_us_per_step": took[1] / CAL * 1e6}
        schedule_note = f"calibrated over {CAL} steps each: serial {took[0] / CAL * 1e6:.1f} us/step, overlapped {took[1] / CAL * 1e6:.1f} us/step"

    def timed_leg(e, first_t, with_events):
        """K steps of env `e` bracketed by barrier + synchronize; returns (elapsed s, [per-launch kernel ms per stream])."""
        if use_dist:
            dist.barrier()
        e.sync()                                                   # device-wide: the group streams are idle too, no stream edges needed
        # HIP events on the stream(s) the kernel is launched on.  groups == 1, N == 1: one pair on torch's current stream
        # (== the stream handed to cda_step) brackets the K back-to-back launches, kernel_ms = span / K.  groups > 1: one
        # pair per group stream, each bracketing that group's K launches.  N > 1 with the all-gather: collectives and
        # stream dependencies sit between launches, so single launches get their own pair - every EV_STRIDE-th one only,
        # a timing event pair costs ~7 us of stream time (tools/host_overhead.py).
        per_launch = gather
        EV_STRIDE = 16
        evs = []
        if with_events and per_launch:
            timed.update({first_t + t: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for t in range(0, K, EV_STRIDE)})
        elif with_events:
            lanes = e.group_streams if e.groups > 1 else [torch.cuda.current_stream(device)]
            if event_lanes == "one":
                lanes = lanes[:1]
            evs = [(s, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for s in lanes]
        t0 = time.perf_counter()
        for s, a, _ in evs:
            a.record(s)
        for t in range(K):
            one_step(e, first_t + t)
        for s, _, b in evs:
            b.record(s)
        e.sync()                                                   # barrier + synchronize on both sides of the K steps (every stream of the device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        if not with_events:
            return elapsed, []
        if per_launch:
            return elapsed, [sum(a.elapsed_time(b) for a, b in timed.values()) / len(timed)]
        return elapsed, [a.elapsed_time(b) / K for _, a, b in evs]

    elapsed, kernel_ms_lanes = timed_leg(env, T0, True)
    flags = env.flags()
    n_flagged = int((flags != 0).sum().item())
    peak_orders = int(env.book_peak().max().item())

    # second leg, one rank only: the same K steps with every info tensor of Info_Helper.set_info (row a14) emitted inside the
    # timed region - the reference's step() always builds info - or, with --info, the info-less leg.
    other = None
    if world == 1 and not args.no_info_leg and not args.fused and not gather:
        env2 = make_env(not args.info)
        for t in range(W):
            one_step(env2, t)
        e2, _ = timed_leg(env2, W, False)
        other = {"elapsed": e2, "flagged": int((env2.flags() != 0).sum().item())}
        env2.close()

    if rank == 0:
        total_agent_steps = float(world) * N * A * K
        value = total_agent_steps / elapsed
        B = ALG_BYTES_CONST + ALG_BYTES_PER_AGENT * A                         # algorithmic bytes per market-step
        markets_per_launch = [c for _, c in env.group_ranges] if env.groups > 1 else [N]
        if len(kernel_ms_lanes) == 1 and len(markets_per_launch) > 1:          # one chain timed: the others run the same launches
            kernel_ms_lanes = kernel_ms_lanes * len(markets_per_launch)
        n_lanes = max(1, len(kernel_ms_lanes))
        # `achieved`: algorithmic bytes of one launch / that launch's duration.  With G concurrent group chains G launches are
        # in flight at any time; the aggregate rate of the device is the sum over the concurrent launches.
        per_launch_gbps = [B * m / (ms * 1e-3) / 1e9 for m, ms in zip(markets_per_launch, kernel_ms_lanes)]
        achieved = sum(per_launch_gbps)
        kernel_ms = sum(kernel_ms_lanes) / n_lanes
        # HBM bytes and issued wave-instructions per market-step from the PMC passes (rocprofv3 --pmc, each counter group in
        # its own run; FETCH_SIZE / WRITE_SIZE calibrated on known byte counts by tools/profile_gpu.sh): committed
        # measurements of THIS workload and build, not live - reported only for the shape they were taken on.
        traffic = traffic_src = issue_frac = valu_busy = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as fh:
                pmc = json.load(fh)
            if (pmc.get("markets"), pmc.get("agents"), bool(pmc.get("info"))) == (N, A, bool(args.info)) and not args.fused and not gather:
                traffic = pmc["hbm_bytes_per_market_step"] * markets_per_launch[0]
                traffic_src = (f"profiles/pmc_latest.json (rocprofv3 --pmc, calibrated; per market-step figure of a {pmc.get('groups')}-chain run "
                               "x markets of one launch)")
                cycles = elapsed / K * GPU_CLOCK_GHZ * 1e9                     # device cycles per step of the whole batch
                issue_frac = pmc["wave_insts_per_market_step"] * N / (N_SIMD * cycles)
                valu_busy = VALU_CYCLES * pmc["valu_insts_per_market_step"] * N / (N_SIMD * cycles)
        except Exception:  # noqa: BLE001
            pass
        out = {
            "metric": "agent-steps/sec (whole node), 4 agents x N parallel markets",
            "value": value, "unit": "agent-steps/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+dec28+f64",
            "data": "synthetic" + ("" if total_steps <= MAX_RESIDENT or args.fused else f" (action stream cycles with period {MAX_RESIDENT} steps)"),
            "config": {"workload": f"{N} markets x {A} random agents per GPU ({'BASELINE configs[2]' if (N, A) == CONFIGS['c3'] else 'per-GPU share of BASELINE configs[3]' if (N, A) == CONFIGS['c4'] else 'custom shape'}); "
                                   f"book pool of 256 resting orders per market shared by both sides (= 128 per side on average; "
                                   f"the reference is unbounded; most held by any market in this run: {peak_orders}); global {world * N} markets"
                                   + (f"; FUSED: {args.fused} steps per launch (cda_run_random)" if args.fused else ""),
                       "markets_per_gpu": N, "agents": A, "info_outputs": bool(args.info), "groups": env.groups,
                       "actions": f"cda_random_actions(seed {ACTION_SEED}, step, global market, agent), resident in HBM",
                       "clock_primer": f"{primer_steps} untimed steps on a scratch env before the measured env's reset",
                       "gather_schedule": schedule_note, "gather_calibration": schedule_us,
                       "collective": ("all_gather(obs|reward|flags slab), " + ("overlapped with the next step on alternating streams" if overlap else "serial")) if gather else "none",
                       "flagged_markets": n_flagged, "peak_resting_orders": peak_orders},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_run_random (per step)" if args.fused else "k_step", "kernel_ms": kernel_ms,
                         "concurrent_launches": n_lanes, "markets_per_launch": markets_per_launch[0],
                         "algorithmic_bytes_per_launch": B * markets_per_launch[0], "achieved_per_launch": per_launch_gbps[0] if per_launch_gbps else None,
                         "issue_frac": issue_frac, "valu_busy_frac": valu_busy},
        }
        if other is not None:
            v2 = total_agent_steps / other["elapsed"]
            key = "without_info" if args.info else "with_info"
            out[f"value_{key}"] = v2
            out[f"ms_per_step_{key}"] = other["elapsed"] / K * 1e3
            out["config"][f"flagged_markets_{key}"] = other["flagged"]
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(N, A, args.cpu_seconds, max_step, first_market)
            except Exception as ex:  # noqa: BLE001 - the baseline is a reported extra, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(out))
    env.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
