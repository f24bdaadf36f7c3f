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

The batch is stepped as `--groups` (default 4; 2 for runs shorter than 200 steps and with the N > 1 hand-back) contiguous market groups, each a chain of
k_step launches on its own stream (cda_step_groups): markets never interact, so the batch-wide barrier of a single launch is
not part of the reference's semantics, and a group's slowest market then overlaps the other groups' work.

What the line reports (one MI355X):
  value / ms_per_step        the HEADLINE leg: step() WITH every info tensor of Info_Helper.set_info (the
                             reference's step() always builds info) inside the timed region, group chains free-running
                             (resident actions, nobody consumes the outputs between steps).  For K < 200 the K-step leg is
                             run `timed_repeats.n` (5) times back to back, each bracketed by barrier + synchronize, and
                             `value` is the MEDIAN leg (min / median / max in `timed_repeats`): a 0.8-ms window is
                             otherwise a coin flip.
  value_without_info         the same with no info tensors (a learner that asks for none)
  value_one_launch           info on, the whole batch as ONE launch per step (--groups 1)
  value_ordered_per_step     info on, the env built with the headline's group chains, every step ordered after the caller's stream and the
                             caller's stream after it (CDAVecEnv.step's default: what a consumer that calls step() per step pays) - since round 6
                             ONE launch of the whole batch on the caller's stream instead of a fork + join of G chains per step
  value_policy_in_loop       a CONSUMER between the steps: every chain runs {policy network forward + action sampling -> env step -> auto reset} for its own
                             markets on its own stream - ONE launch per step where the env qualifies (k_policy_step: the policy evaluated inside the step
                             kernel, include/cda.h cda_policy_step_range), else a hand-written MFMA launch + the step launch -, K steps per HIP graph, no
                             cross-stream edge inside the K steps (mlp.RolloutChains: the rollout of the PPO loop of BASELINE configs[4], random
                             initial weights, no info tensors) - what a learner in the loop gets, where value_ordered_per_step is what it
                             would get through per-step event edges
  value_run_random_one_launch  row H (CDA_rand.run_random): 256 steps of uniform random agents for every market in ONE launch (cda_run_random) - what the batch does when
                             no market-wave ever waits for the batch's slowest one (a per-step launch lasts as long as its slowest wave)
  value_league_self_play     the reference's training topology END TO END on the fused kernels (league_train.train_league_fused): 8192 markets x 8 agents, 2
                             separately trained policies against random modules + champion snapshots, rollout + both PPO updates per iteration
  roofline                   HIP event pairs on the chains' own streams around the k_step launches of the headline leg;
                             `traffic` / `issue_frac` only when a committed PMC pass of exactly this shape exists
                             (profiles/pmc/<markets>x<agents>_info<0|1>_g<groups>.json)
N > 1 (one rank per GPU): every chain also hands its markets' compact records (newest frame | reward | flags) to all ranks with
its OWN all-gather on its OWN stream and communicator and rebuilds the learner-side arrays there (parallel.py); the timed
region contains all of it.

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
    p.add_argument("--groups", type=int, default=None, help="concurrent market groups per GPU (default 4, 2 below 200 timed steps)")
    p.add_argument("--event-lanes", choices=["all", "one"], default=None,
                   help="HIP event pairs on every group stream, or on the first one only (default: all, one below 200 timed steps)")
    p.add_argument("--repeats", type=int, default=None, help="timed K-step legs run back to back (default 5 below 200 timed steps, else 1); value = the median leg")
    p.add_argument("--no-info", action="store_true", help="headline leg WITHOUT the info tensors (then value_with_info is the extra)")
    p.add_argument("--no-extra-legs", action="store_true", help="skip value_without_info / value_one_launch / value_ordered_per_step")
    p.add_argument("--no-gather", action="store_true", help="N>1: skip the hand-back (records all-gather + rebuild)")
    p.add_argument("--force-gather", action="store_true",
                   help="run the N>1 code path (process group, per-chain all-gather, rebuild) even with one rank; diagnostics")
    p.add_argument("--transport", choices=["auto", "rccl", "torch"], default=os.environ.get("CDA_BENCH_TRANSPORT", "auto"),
                   help="N>1 hand-back: 'rccl' = ncclAllGather issued natively per chain (cda_step_groups_handback), 'torch' = torch.distributed collectives per chain, "
                        "'auto' = rccl when the process group runs on RCCL and the start-up self-check passes (env: CDA_BENCH_TRANSPORT)")
    p.add_argument("--probe-native", action="store_true", help="internal: a sacrificial child of an N > 1 run - steps a tiny env through the native RCCL hand-back and exits 0 only if "
                                                               "that transport carried it (see probe_native_transport)")
    p.add_argument("--no-policy-leg", action="store_true", help="skip value_policy_in_loop")
    p.add_argument("--no-league-leg", action="store_true", help="skip value_league_self_play")
    p.add_argument("--fused", type=int, default=0, metavar="T",
                   help="not the headline run: T steps per launch through cda_run_random (random agents sampled in the kernel, "
                        "market state resident in LDS across steps, no per-step barrier between markets)")
    p.add_argument("--learner", choices=["none", "dp"], default="none",
                   help="not the headline run: 'dp' = the PPO loop of BASELINE configs[4] (ppo.train_fused, hand-written network kernels) as a DATA-PARALLEL learner - every rank "
                        "rolls out and back-propagates its own shard of markets, the ranks all-reduce the 0.9-MB gradient once per minibatch step (SURVEY 8(e): 'if the learner is "
                        "itself data-parallel over the same shards, the all-gather can be skipped entirely'); here a step = one iteration (a 64-step rollout + its update)")
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


def physical_cores():
    try:
        ids = set()
        with open("/proc/cpuinfo") as fh:
            phys = core = None
            for ln in fh:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        ids.add((phys, core))
                    phys = core = None
        return len(ids) or "?"
    except OSError:
        return "?"


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

    def timed(n_markets, n_threads, steps, with_info=True):
        env = O.OracleEnv(cfg, n_markets=n_markets)
        env.reset(seeds=seeds[:n_markets])
        bounds = [(i * n_markets // n_threads, (i + 1) * n_markets // n_threads) for i in range(n_threads)]
        bounds = [(lo, hi) for lo, hi in bounds if hi > lo]
        import ctypes as C
        info = C.byref(env._info_ptrs) if with_info else None          # every info tensor of Info_Helper.set_info, like the GPU headline leg

        def work(lo, hi):      # ONE foreign call per thread; ctypes releases the GIL for its duration
            lib.oracle_run_random_range_info(env.h, lo, hi - lo, 0, steps, ACTION_SEED, first_market, env.obs.ctypes.data, env.reward.ctypes.data,
                                             env.term.ctypes.data, env.trunc.ctypes.data, info)
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
    phys = physical_cores()
    dt, _ = timed(n, threads, 8)                                   # calibration on a throw-away env
    steps = int(max(16, min(max_step - 1, 0.55 * budget_s / max(dt / 8, 1e-6))))
    dt, used = timed(n, threads, steps)                            # steps 0 .. steps-1 of the GPU leg's stream
    value = n * agents * steps / dt
    # one thread per PHYSICAL core (the SMT siblings idle): the figure a core count should be read against
    value_phys = None
    if isinstance(phys, int) and 0 < phys < threads:
        stp = int(max(8, steps * 0.35))
        dtp, usedp = timed(n, phys, stp)
        value_phys = n * agents * stp / dtp
    n1 = min(n, 64)
    dt1, _ = timed(n1, 1, 4)
    steps1 = int(max(8, min(max_step - 1, 0.15 * budget_s / max(dt1 / 4, 1e-6))))
    dt1, _ = timed(n1, 1, steps1)
    value1 = n1 * agents * steps1 / dt1
    dtn, _ = timed(n, threads, max(8, steps // 4), with_info=False)
    value_noinfo = n * agents * max(8, steps // 4) / dtn
    return {"value": value, "unit": "agent-steps/s", "cores": used, "cores_note": f"{used} hardware threads ({phys} physical cores, SMT)",
            "kind": "port", "outputs": "obs + reward + flags + every info tensor (as the GPU headline leg)",
            "value_physical_cores": value_phys, "physical_cores": phys if isinstance(phys, int) else None,
            "value_without_info": value_noinfo, "value_1thread": value1, "cpu_model": cpu_model(),
            "sample": f"{n} markets x {agents} agents x steps 0..{steps - 1} of the GPU leg's action stream ({dt:.1f} s wall), C oracle, "
                      f"{used} threads, info tensors ON; one thread per physical core ({phys}): {'%.3g' % value_phys if value_phys else 'n/a'} agent-steps/s; "
                      f"without info tensors: {value_noinfo:.3g}; 1 thread: {n1} markets x {steps1} steps ({dt1:.1f} s)"}


def pmc_entry(n, a, info, groups):
    """A committed PMC pass (tools/profile_gpu.sh -> profiles/pmc/...) of EXACTLY this shape, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc", f"{n}x{a}_info{int(bool(info))}_g{groups}.json")) as fh:
            pmc = json.load(fh)
        if (pmc.get("markets"), pmc.get("agents"), bool(pmc.get("info")), pmc.get("groups")) == (n, a, bool(info), groups):
            return pmc
    except Exception:  # noqa: BLE001
        pass
    return None


def probe_native_transport(dist, world, rank, local_rank, device, timeout_s=150.0):
    """The native hand-back (ncclAllGather issued by the library itself, one communicator per chain) has never run with more than one real RCCL rank: no multi-GPU node was
    available while it was built.  A collective that hangs cannot be recovered from inside the process that issued it (its kernels spin on the device), so before an
    N > 1 run trusts it, every rank sends a SACRIFICIAL CHILD process through it: `bench.py --probe-native` on this rank's GPU, a rendezvous of its own (MASTER_PORT + 23),
    64 markets, a few steps, transport forced to rccl.  The child exits 0 only if the native transport carried the records and the learner-side arrays agreed with
    torch.distributed's (ShardedVecEnv's start-up self-check); a child that hangs is killed after `timeout_s` - with it its device work - and costs the parent nothing.
    The ranks then agree (all-reduce MIN): native for all, or torch.distributed for all.  Returns (use_native, note)."""
    import subprocess
    import torch
    # (TORCHELASTIC_*: under torch.distributed.run the agent hosts the rendezvous store and ranks only connect to it - the children have no agent: rank 0's must host its own)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + 23)
    env["RANK"], env["WORLD_SIZE"], env["LOCAL_RANK"] = str(rank), str(world), str(local_rank)
    env["CDA_HANDBACK_TIMEOUT_S"] = str(min(60.0, timeout_s / 2))
    cmd = [sys.executable, os.path.abspath(__file__), "--probe-native", "--gpus", str(world), "--markets", "64", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--no-extra-legs", "--transport", "rccl"] + (["--force-gather"] if world == 1 else [])     # (one rank: a one-rank communicator, so that one GPU runs the real call)
    if world > 1:
        dist.barrier()                                               # the children rendezvous with each other: start them together
    ok, note = 0.0, ""
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        ok = 1.0 if out.returncode == 0 else 0.0
        if not ok:
            note = f"rank {rank}: the probe exited {out.returncode}: {(out.stderr or out.stdout).strip().splitlines()[-1][:200] if (out.stderr or out.stdout).strip() else ''}"
    except subprocess.TimeoutExpired:
        note = f"rank {rank}: the probe did not finish within {timeout_s:.0f} s (killed)"
    t = torch.tensor([ok], device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    use = float(t.item()) >= 1.0
    return use, ("the native RCCL hand-back passed a sacrificial-child probe on every rank" if use else
                 f"the native RCCL hand-back failed its sacrificial-child probe ({note or 'on another rank'}): torch.distributed carries the records")


def learner_dp(args, dist, world, rank, device, backend):
    """--learner dp: W + K iterations of ppo.train_fused on this rank's shard, gradients summed over the ranks; K iterations timed (max over ranks of the
    per-iteration device times, each bracketed by synchronize).  One JSON line on rank 0."""
    import torch
    from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
    from gym_continuousdoubleauction_amd.parallel import make_grad_allreduce
    N, A, K, W, T = learner_shape(args)
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=N, device=str(device), with_info=False)
    allreduce = make_grad_allreduce(dist) if world > 1 else None
    _, hist = ppo.train_fused(env, iters=W + K, horizon=T, seed=0, log=lambda s: None, chains=args.groups or 4, allreduce=allreduce, world=world, first_market=rank * N)
    dt = torch.tensor([sum(h["rollout_s"] + h["update_s"] for h in hist[W:])], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    flags = int((env.flags() != 0).sum().item())
    env.close()
    if rank == 0:
        elapsed = float(dt.item())
        print(json.dumps(learner_line(world, N, A, T, K, W, elapsed, backend, flags, {k: hist[-1][k] for k in ("pg_loss", "v_loss", "entropy")})))
    if world > 1:
        dist.destroy_process_group()


def run_shape(args, use_dist):
    """Shape of the headline run from the command line: (N markets per GPU, A agents, K timed steps, W warm-up steps, R timed legs, hand-back?, chains, event lanes,
    total steps, max_step)."""
    N, A = CONFIGS[args.config]
    N = args.markets if args.markets is not None else N
    A = args.agents if args.agents is not None else A
    K, W = args.steps, args.warmup
    if args.fused and (args.steps % args.fused or args.warmup % args.fused):
        raise SystemExit("--fused T needs --steps and --warmup to be multiples of T")
    gather = use_dist and not args.no_gather and not args.fused
    R = max(1, args.repeats if args.repeats is not None else (5 if K < 200 and not args.fused else 1))
    total_steps = W + R * K
    max_step = max(4096, total_steps + 1)                          # no truncation inside the run
    # default number of group chains: 4 pays once the chains are long, 2 when the whole timed region is a few dozen steps and the
    # staggered start / drain of four chains is a visible share of it (profiles/r02)
    event_lanes = args.event_lanes or ("all" if K >= 200 else "one")
    # with the hand-back two chains: every chain adds a collective and a rebuild launch per step to the host's work, and at four chains
    # the host, not the GPU, bounds the step (tools/handback_rccl_probe.py: 47 us per step at four chains, 41.6 us at two)
    groups = args.groups if args.groups is not None else (1 if args.fused else (2 if gather else (4 if K >= 200 else 2)))
    groups = max(1, min(groups, N))
    return N, A, K, W, R, gather, groups, event_lanes, total_steps, max_step


def learner_shape(args):
    """--learner dp: (N, A, K timed iterations, W warm-up iterations, T steps per rollout)."""
    N, A = CONFIGS[args.config]
    N = args.markets if args.markets is not None else N
    A = args.agents if args.agents is not None else A
    return N, A, (args.steps if args.steps != 1000 else 8), (args.warmup if args.warmup != 64 else 2), 64


def learner_line(world, N, A, T, K, W, elapsed, backend, flags, losses):
    """The ONE JSON line of `--learner dp` (no device work in here; see headline_line)."""
    return {"metric": "agent-steps/sec end to end (rollout + PPO update), data-parallel learner over the market shards", "value": world * N * A * T * K / elapsed,
            "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 operands, f32 accumulation (network); int32+dec28+f64 (env)", "data": "synthetic",
            "config": {"workload": f"{N} markets x {A} agents per GPU (global {world * N}), PPO policy on the hand-written MFMA network kernels in the loop, horizon {T}, "
                                   "4 epochs, 262144-sample minibatches; a step = one iteration",
                       "collective": ("none (one rank)" if world == 1 else
                                      f"one all-reduce of the 0.9-MB gradient per minibatch step + one of the two advantage sums per rollout, torch.distributed over {backend}; "
                                      "no observation / reward hand-back"),
                       "flagged_markets": flags},
            "loss_last_iteration": losses}


def group_ranges(n, groups):
    """[(first market, count)] of the contiguous market groups: the library's own rule (cda_group_range, a host function)."""
    import ctypes as C
    from gym_continuousdoubleauction_amd._lib import lib
    out = []
    for g in range(groups):
        first, cnt = C.c_int32(), C.c_int32()
        lib().cda_group_range(n, groups, g, C.byref(first), C.byref(cnt))
        out.append((first.value, cnt.value))
    return out


def dry_run(args):
    """CDA_BENCH_DRY_RUN=1 (tests/test_bench_dry_contract.py; no GPU, no library): everything of an N-rank run that is NOT device work - the rendezvous from the
    launcher's environment (gloo), the barriers, the MAX-over-ranks reduction of the elapsed time, the shape logic of the command line and the line's schema -
    with FABRICATED measurements (rank r 'measures' 40 us x (1 + r / 100) per step), so that the first real multi-GPU run cannot fail on bookkeeping.
    The line says `"data": "dry run"`; nothing in it is a measurement."""
    import types
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    use_dist = world > 1 or args.force_gather
    backend = "gloo"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=PG_TIMEOUT)

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64)
        if use_dist:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.learner == "dp":
        N, A, K, W, T = learner_shape(args)
        elapsed = max_over_ranks(K * 7e-3 * (1 + rank / 100))
        out = learner_line(world, N, A, T, K, W, elapsed, backend, 0, {"pg_loss": 0.0, "v_loss": 0.0, "entropy": 0.0})
    else:
        N, A, K, W, R, gather, groups, event_lanes, total_steps, max_step = run_shape(args, use_dist)
        took = [max_over_ranks(K * 40e-6 * (1 + rank / 100)) for _ in range(R)]
        transport = "torch" if args.transport == "auto" else args.transport
        out = headline_line(args, types.SimpleNamespace(
            world=world, n_gpus=world, N=N, A=A, K=K, W=W, R=R, took=took, kms=[[0.036]] * R, head_ranges=group_ranges(N, groups), head_groups=groups,
            headline_info=not args.no_info, gather=gather, tile=0, spill=0, spill_wanted=0, peak_orders=0, primer_steps=0,
            head_transport=None if not gather else ("ncclAllGather issued by cda_step_groups_handback" if transport == "rccl" else "torch.distributed"),
            head_transport_note="dry run" if gather else None, n_flagged=0, total_steps=total_steps, MAX_RESIDENT=4096, policy_leg=None, league_leg=None, rr_leg=None, extras={}))
    out["data"] = "dry run (fabricated timings: CDA_BENCH_DRY_RUN=1)"
    if "roofline" in out:
        out["roofline"].update(traffic=None, traffic_source=None, issue_frac=None, valu_busy_frac=None)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def headline_line(args, m):
    """The ONE JSON line of the headline run from its measurements `m` (a namespace: the timed legs' wall times, the HIP-event kernel times, the shape, the extra legs).
    No device work in here: tests/test_bench_dry_contract.py builds the same line from fabricated measurements under a world-2 gloo group."""
    import statistics
    from gym_continuousdoubleauction_amd.parallel import handback_stride
    world = m.world
    n_gpus = m.n_gpus
    N = m.N
    A = m.A
    K = m.K
    W = m.W
    R = m.R
    took = m.took
    kms = m.kms
    head_ranges = m.head_ranges
    head_groups = m.head_groups
    headline_info = m.headline_info
    gather = m.gather
    tile = m.tile
    spill = m.spill
    spill_wanted = m.spill_wanted
    peak_orders = m.peak_orders
    primer_steps = m.primer_steps
    head_transport = m.head_transport
    head_transport_note = m.head_transport_note
    n_flagged = m.n_flagged
    total_steps = m.total_steps
    MAX_RESIDENT = m.MAX_RESIDENT
    policy_leg = m.policy_leg
    league_leg = m.league_leg
    rr_leg = m.rr_leg
    extras = m.extras
    total_agent_steps = float(world) * N * A * K
    elapsed = statistics.median(took)
    value = total_agent_steps / elapsed
    B = ALG_BYTES_CONST + ALG_BYTES_PER_AGENT * A                         # algorithmic bytes per market-step
    markets_per_launch = [c for _, c in head_ranges] if head_groups > 1 else [N]
    lanes_ms = [statistics.mean(x) for x in zip(*kms)] if kms else []     # per timed stream, averaged over the repeats
    if len(lanes_ms) == 1 and len(markets_per_launch) > 1:                 # one chain timed: the others run the same launches
        lanes_ms = lanes_ms * len(markets_per_launch)
    n_lanes = max(1, len(lanes_ms))
    # `achieved`: algorithmic bytes of one launch / that launch's duration.  With G concurrent group chains G launches are
    # in flight at any time; the aggregate rate of the device is the sum over the concurrent launches.
    per_launch_gbps = [B * m / (ms * 1e-3) / 1e9 for m, ms in zip(markets_per_launch, lanes_ms)]
    achieved = sum(per_launch_gbps)
    kernel_ms = sum(lanes_ms) / n_lanes if lanes_ms else None
    # HBM bytes and issued wave-instructions per market-step from committed PMC passes of EXACTLY this shape (markets, agents,
    # info, chains) - rocprofv3 --pmc, each counter group in its own run, FETCH_SIZE / WRITE_SIZE calibrated on known byte
    # counts (tools/profile_gpu.sh); null otherwise.
    traffic = traffic_src = issue_frac = valu_busy = None
    pmc = None if (args.fused or gather) else pmc_entry(N, A, headline_info, head_groups)
    if pmc:
        traffic = pmc["hbm_bytes_per_market_step"] * markets_per_launch[0]
        traffic_src = f"profiles/pmc/{N}x{A}_info{int(headline_info)}_g{head_groups}.json (rocprofv3 --pmc, calibrated; this build, this shape)"
        cycles = elapsed / K * GPU_CLOCK_GHZ * 1e9                         # device cycles per step of the whole batch
        issue_frac = pmc["wave_insts_per_market_step"] * N / (N_SIMD * cycles)
        valu_busy = VALU_CYCLES * pmc["valu_insts_per_market_step"] * N / (N_SIMD * cycles)
    shape = ("BASELINE configs[2]" if (N, A) == CONFIGS["c3"] else "per-GPU share of BASELINE configs[3]" if (N, A) == CONFIGS["c4"] else "custom shape")
    how = "FUSED episodes (cda_run_random)" if args.fused else (
        f"step() {'with every info tensor (row a14)' if headline_info else 'without info tensors'}, {head_groups} free-running group chain(s)"
        + (", per-chain hand-back of the new frame | reward | flags to every rank" if gather else ""))
    out = {
        "metric": f"agent-steps/sec (whole node), 4 agents x N parallel markets; {how}",
        "value": value, "unit": "agent-steps/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32+dec28+f64",
        "data": "synthetic" + ("" if total_steps <= MAX_RESIDENT or args.fused else f" (action stream cycles with period {MAX_RESIDENT} steps)"),
        "timed_repeats": {"n": R, "ms_per_step": [x / K * 1e3 for x in took], "min": min(took) / K * 1e3, "median": elapsed / K * 1e3,
                          "max": max(took) / K * 1e3, "value_is": "the median leg" if R > 1 else "the one leg"},
        "config": {"workload": f"{N} markets x {A} random agents per GPU ({shape}); unbounded book: LDS tile of {tile} resting orders per market "
                               f"(the top of the book, both sides) + HBM spill ring of {spill} per side (the reference's OrderTree is unbounded; "
                               f"most held by any market in this run: {peak_orders}); global {world * N} markets"
                               + (f"; FUSED: {args.fused} steps per launch (cda_run_random)" if args.fused else ""),
                   "markets_per_gpu": N, "agents": A, "info_outputs": bool(headline_info), "groups": head_groups,
                   "book_tile": tile, "book_spill": spill, "book_spill_wanted": spill_wanted, "spill_halved": bool(spill < spill_wanted),
                   "actions": f"cda_random_actions(seed {ACTION_SEED}, step, global market, agent), resident in HBM",
                   "clock_primer": f"{primer_steps} untimed steps on a scratch env before the measured envs' resets",
                   "collective": (f"{head_groups} all-gathers per step (one per chain, own stream + communicator; transport: {head_transport}"
                                  + (f" [{head_transport_note}]" if head_transport_note else "") + f"; asked: {args.transport}) of {handback_stride(A)}-B records "
                                  f"(newest frame | reward | flags), rebuilt into [global markets, ...] arrays by cda_handback_unpack") if gather else "none",
                   "flagged_markets": n_flagged, "peak_resting_orders": peak_orders},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k_run_random (per step)" if args.fused else "k_step", "kernel_ms": kernel_ms,
                     "concurrent_launches": n_lanes, "markets_per_launch": markets_per_launch[0],
                     "algorithmic_bytes_per_launch": B * markets_per_launch[0], "achieved_per_launch": per_launch_gbps[0] if per_launch_gbps else None,
                     "issue_frac": issue_frac, "valu_busy_frac": valu_busy},
    }
    if policy_leg is not None:
        if "error" in policy_leg:
            out["value_policy_in_loop"] = None
            out["config"]["policy_in_loop"] = f"failed: {policy_leg['error']}"
        else:
            out["value_policy_in_loop"] = total_agent_steps / policy_leg["elapsed"]
            out["ms_per_step_policy_in_loop"] = policy_leg["elapsed"] / K * 1e3
            out["config"]["policy_in_loop"] = (f"{policy_leg['chains']} chains x {{168-512-512-32 bf16 MFMA policy/value forward + sampling -> k_step -> auto reset}}, "
                                               f"{K} steps per HIP graph" + ("" if policy_leg["graphs"] else " (graph capture failed: direct launches)")
                                               + f", no info tensors; min / max over the repeats: {total_agent_steps / policy_leg['max']:.4g} / {total_agent_steps / policy_leg['min']:.4g}")
            out["config"]["flagged_markets_policy_in_loop"] = policy_leg["flagged"]
            if "metrics_on" in policy_leg:
                out["value_policy_in_loop_episode_metrics"] = total_agent_steps / policy_leg["metrics_on"]
                out["config"]["policy_in_loop_episode_metrics"] = (
                    "the same loop over 64-step episodes with every episode checked (exact sum of NAV) and summarised on the device in the in-kernel auto reset, the step "
                    f"tallying the callback's counters and reward terms, + one collection per rollout: {total_agent_steps / policy_leg['metrics_on']:.4g} against "
                    f"{total_agent_steps / policy_leg['metrics_off']:.4g} agent-steps/s with the metrics off on the same env ({(policy_leg['metrics_on'] / policy_leg['metrics_off'] - 1) * 100:+.2f} % time); "
                    f"last collection: {policy_leg['metrics_episodes']:.0f} episodes, {policy_leg['metrics_violations']:.0f} violations")
    if league_leg is not None:
        if "error" in league_leg:
            out["value_league_self_play"] = None
            out["config"]["league_self_play"] = f"failed: {league_leg['error']}"
        else:
            out["value_league_self_play"] = league_leg["value"]
            lN, lA, lT = league_leg["shape"]
            out["config"]["league_self_play"] = (f"end to end (rollout + one PPO update per trainable policy): {lN} markets x {lA} agents, 2 separately trained policies against "
                                                 f"6 uniform random modules + champion snapshots drawn per episode and slot by the reference's mapping rule (on the device), {lT}-step "
                                                 f"episodes; {league_leg['timed']} of {league_leg['iterations']} iterations timed: rollout {league_leg['rollout_ms']:.2f} ms + updates "
                                                 f"{league_leg['update_ms']:.2f} ms per iteration, {league_leg['champions']} champions promoted, {league_leg['flagged']} flagged markets")
    if rr_leg is not None:
        out["value_run_random_one_launch"] = rr_leg.get("value")
        out["config"]["run_random_one_launch"] = (f"failed: {rr_leg['error']}" if "error" in rr_leg else
                                                  f"cda_run_random: {rr_leg['steps']} steps of uniform random agents for every market in one launch ({rr_leg['ms']:.2f} ms; in-kernel "
                                                  f"counter-based sampler, no info tensors, {rr_leg['flagged']} flagged markets) - the reference's CDA_rand.run_random, SURVEY 8 row H")
    for name, ex in extras.items():
        out[f"value_{name}"] = total_agent_steps / ex["elapsed"]
        out[f"ms_per_step_{name}"] = ex["elapsed"] / K * 1e3
        out["config"][f"flagged_markets_{name}"] = ex["flagged"]
    return out


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
    if os.environ.get("CDA_BENCH_DRY_RUN") == "1":
        return dry_run(args)
    import statistics

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
    from gym_continuousdoubleauction_amd.parallel import ShardedVecEnv, handback_stride

    if args.learner == "dp":
        return learner_dp(args, dist, world, rank, device, backend)
    probe_note = None
    if world > 1 and backend == "nccl" and args.transport == "auto" and not args.no_gather and not args.fused and not args.probe_native \
            and os.environ.get("CDA_BENCH_PROBE_NATIVE", "1") != "0":
        use_native, probe_note = probe_native_transport(dist, world, rank, local_rank, device)
        args.transport = "rccl" if use_native else "torch"
    N, A, K, W, R, gather, groups, event_lanes, total_steps, max_step = run_shape(args, use_dist)
    cfg = {"num_of_agents": A, "init_cash": 1000000, "max_step": max_step, "is_render": False}
    first_market = rank * N                           # global market index -> seed and action key, independent of the GPU count
    seeds = (SEED_BASE + first_market + torch.arange(N, dtype=torch.int64)).numpy().astype("uint64")
    headline_info = not args.no_info

    class Leg:
        """One measured env: `groups` chains, info on / off, optionally the N > 1 hand-back, free-running or ordered per step."""

        def __init__(self, with_info, n_groups, handback=False, ordered=False):
            self.with_info, self.ordered, self.handback = with_info, ordered, handback
            if handback:
                self.sh = ShardedVecEnv(cfg, world * N, device=str(device), groups=n_groups, handback=True, force_collective=args.force_gather, transport=args.transport,
                                        env_factory=lambda c, n, d, g: CDAVecEnv(c, n_markets=n, device=d, with_info=with_info, groups=g, handback=True))
                self.env = self.sh.env
                self.sh.reset(seed_base=SEED_BASE)
            else:
                self.sh = None
                self.env = CDAVecEnv(cfg, n_markets=N, device=str(device), with_info=with_info, groups=n_groups)
                self.env.reset(seed=seeds)

        def step(self, t):
            i = t % period
            if args.fused:
                if t % args.fused == 0:
                    self.env.run_random(args.fused, action_seed=ACTION_SEED, market_index_base=first_market)
            elif self.sh is not None:
                self.sh.step(*step_acts[i], pipelined=True)
            else:
                self.env.step(*step_acts[i], pipelined=not self.ordered)

        def close(self):
            (self.sh or self.env).close()

    # Clock primer: an idle MI355X sits in a low-power state and needs tens of milliseconds of work before its engine clock
    # is up; the driver's default run times 20 steps (~1 ms) right after start-up, which would measure the ramp, not the
    # kernel.  A SCRATCH env (not a measured one) is stepped for ~PRIMER_MS first; every measured env then does exactly
    # W untimed + R x K timed steps from its own reset.
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
    # The action stream of the whole run, resident in HBM (20 B per agent-step: 350 MB for the default 1064 steps of
    # 4096 x 4); beyond MAX_RESIDENT steps the run cycles through the first MAX_RESIDENT (reported in `data`).
    MAX_RESIDENT = 4096
    period = min(total_steps, MAX_RESIDENT)
    acts = None
    if not args.fused:
        gen = CDAVecEnv(cfg, n_markets=N, device=str(device), with_info=False)
        acts = gen.random_actions_device(0, period, action_seed=ACTION_SEED, market_index_base=first_market)
        torch.cuda.synchronize()
        gen.close()
        # step t's five [N, A] tensors as views made ONCE, outside the timed region (indexing five tensors per step costs the host ~8 us, a fifth of a step)
        step_acts = [tuple(a[i] for a in acts) for i in range(period)]

    def run_leg(leg, with_events):
        """W untimed steps, then R legs of EXACTLY K steps, each bracketed by barrier + synchronize on both sides (max over ranks).
        Returns ([elapsed s per leg], [kernel ms per launch per timed stream])."""
        e = leg.env
        for t in range(W):
            leg.step(t)
        lanes = (e.group_streams if e.groups > 1 else [torch.cuda.current_stream(device)])
        if event_lanes == "one":
            lanes = lanes[:1]
        # HIP events on the stream(s) the kernel is launched on.  Without the hand-back one pair per stream brackets that chain's K
        # back-to-back launches (kernel_ms = span / K).  With it, collectives and the rebuild sit between a chain's launches, so
        # single launches of the first chain get their own pair - every EV_STRIDE-th step only, a pair costs ~7 us of stream time.
        EV_STRIDE = 16
        took, kms = [], []
        for r in range(R):
            base = W + r * K
            spans, singles = [], []
            if use_dist:
                dist.barrier()
            e.sync()                                               # device-wide: every stream is idle, no stream edges needed
            if with_events and not leg.handback:
                spans = [(s, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for s in lanes]
            t0 = time.perf_counter()
            for s, a, _ in spans:
                a.record(s)
            for t in range(K):
                if with_events and leg.handback and t % EV_STRIDE == 0:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(lanes[0])
                    e.step(*step_acts[(base + t) % period], pipelined=True)
                    b.record(lanes[0])
                    leg.sh.handback()
                    singles.append((a, b))
                else:
                    leg.step(base + t)
            for s, _, b in spans:
                b.record(s)
            e.sync()                                               # barrier + synchronize on both sides of the K steps (every stream of the device)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if use_dist:
                tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elapsed = float(tt.item())
            took.append(elapsed)
            if spans:
                kms.append([a.elapsed_time(b) / K for _, a, b in spans])
            elif singles:
                kms.append([sum(a.elapsed_time(b) for a, b in singles) / len(singles)])
        return took, kms

    torch.cuda.synchronize()
    head = Leg(headline_info, groups, handback=gather)
    took, kms = run_leg(head, True)
    env = head.env
    n_flagged = int((env.flags() != 0).sum().item())
    peak_orders = int(env.book_peak().max().item())
    tile, spill, spill_wanted = env.book_capacity, env.book_spill, env.book_spill_wanted
    head_groups, head_ranges = env.groups, list(env.group_ranges)
    head_transport = None if head.sh is None else ("ncclAllGather issued by cda_step_groups_handback" if head.sh.transport == "rccl" else "torch.distributed")
    head_transport_note = None if head.sh is None else getattr(head.sh, "transport_note", None)
    if probe_note:
        head_transport_note = f"{probe_note}; {head_transport_note}" if head_transport_note else probe_note
    if args.probe_native:                                            # the sacrificial child: its exit code is the verdict (the legs above went through the transport)
        ok = head.sh is not None and head.sh.transport == "rccl" and n_flagged == 0
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(); dist.destroy_process_group()
        sys.exit(0 if ok else 4)
    head.close()

    extras = {}
    if world == 1 and not args.no_extra_legs and not args.fused and not gather:
        def extra(name, **kw):
            leg = Leg(**kw)
            tk, _ = run_leg(leg, False)
            extras[name] = {"elapsed": statistics.median(tk), "flagged": int((leg.env.flags() != 0).sum().item())}
            leg.close()
        extra("with_info" if not headline_info else "without_info", with_info=not headline_info, n_groups=groups)
        if groups != 1:
            extra("one_launch", with_info=headline_info, n_groups=1)
            extra("ordered_per_step", with_info=headline_info, n_groups=groups, ordered=True)

    # the consumer-facing leg: policy network in the loop, per chain (see the module docstring)
    policy_leg = None
    if world == 1 and not args.no_extra_legs and not args.no_policy_leg and not args.fused and not gather and cfg.get("n_hist", 4) == 4:
        try:
            from gym_continuousdoubleauction_amd.mlp import FusedPolicy, RolloutChains
            pcfg = dict(cfg, auto_reset=True)
            penv = CDAVecEnv(pcfg, n_markets=N, device=str(device), with_info=False)
            penv.reset(seed=seeds)
            pol = FusedPolicy(device, seed=0)
            # chains: 4 market groups on 4 streams for long rollouts (their launches overlap each other's tails: +3 % at 256 steps), ONE chain on the caller's stream below
            # 200 steps - every further chain starts ~35 us after its predecessor (one hipGraphLaunch each) and the fork / join edges cost 30-50 us, which a 1-ms
            # rollout does not earn back (tools/policy_leg_probe.py, profiles/r06/policy_leg_probe_*.jsonl; the headline leg makes the same choice with --groups)
            chains = max(1, min(4, N)) if K >= 200 else 1
            roll = RolloutChains(penv, pol, K, groups=chains, seed=ACTION_SEED, use_graphs=True)
            for _ in range(max(2, min(4, W // max(K, 1) + 2))):        # warm-up rollouts (the first one captures the chains' graphs)
                roll.run()
            ptook = []
            for _ in range(max(R, 3)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                roll.run()
                torch.cuda.synchronize()
                ptook.append(time.perf_counter() - t0)
            policy_leg = {"elapsed": statistics.median(ptook), "min": min(ptook), "max": max(ptook), "chains": chains, "graphs": roll.graphs is not None,
                          "flagged": int((penv.flags() != 0).sum().item())}
            penv.close()
            del roll
            # ... and the same loop with EVERY EPISODE CHECKED AND SUMMARISED ON THE DEVICE (cda_episode_metrics_enable): per-step tallies of what the reference's callback
            # tallies, sum-of-NAV conservation + the episode summary in the in-kernel auto reset; short episodes (64 steps) so that ends happen inside the timed rollouts
            menv = CDAVecEnv(dict(pcfg, max_step=64), n_markets=N, device=str(device), with_info=False)
            menv.reset(seed=seeds)
            legs = {}
            for on in (False, True):
                menv.enable_episode_metrics(on)
                mroll = RolloutChains(menv, pol, K, groups=chains, seed=ACTION_SEED, use_graphs=True)
                for _ in range(3):
                    mroll.run()
                mtook = []
                for _ in range(max(R, 5)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    mroll.run()
                    if on:
                        mtab = menv.collect_episode_metrics()
                    torch.cuda.synchronize()
                    mtook.append(time.perf_counter() - t0)
                legs[on] = statistics.median(mtook)
                del mroll
            policy_leg.update(metrics_on=legs[True], metrics_off=legs[False], metrics_episodes=float(mtab[1][0].item()), metrics_violations=float(mtab[1][1].item()))
            menv.close()
            del pol
        except Exception as ex:  # noqa: BLE001 - an extra leg never fails the headline
            policy_leg = {"error": repr(ex)}

    # row H (the reference's CDA_rand.run_random): whole random-agent episodes in ONE launch per batch - a market-wave never waits for the batch's slowest wave
    rr_leg = None
    if world == 1 and not args.no_extra_legs and not args.fused and not gather:
        try:
            renv = CDAVecEnv(dict(cfg, max_step=1 << 20), n_markets=N, device=str(device), with_info=False)
            renv.reset(seed=seeds)
            rsteps = 256
            renv.run_random(rsteps, action_seed=ACTION_SEED)                   # warm-up launch
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            _, _, _, _, taken = renv.run_random(rsteps, action_seed=ACTION_SEED + 1)
            ev1.record()
            torch.cuda.synchronize()
            rr_leg = {"value": float(taken.sum().item()) * A / (ev0.elapsed_time(ev1) * 1e-3), "steps": rsteps, "ms": ev0.elapsed_time(ev1), "flagged": int((renv.flags() != 0).sum().item())}
            renv.close()
        except Exception as ex:  # noqa: BLE001
            rr_leg = {"error": repr(ex)}

    # the reference's own training topology, end to end (league_train.train_league_fused): 2048 markets x 8 agents, 2 separately trained policies against random
    # modules + champion snapshots, rollout + both updates per iteration - a learner-side extra, reported in agent-steps/s like everything else on the line
    league_leg = None
    if world == 1 and not args.no_extra_legs and not args.no_policy_leg and not args.no_league_leg and not args.fused and not gather:
        try:
            from gym_continuousdoubleauction_amd.league_train import train_league_fused
            # the reference fixes 8 agents / 2 trainable policies / the pool's weights, not the number of parallel games: 8192 markets fill the machine (two rounds of
            # market-waves per launch; profiles/r06/bench_league_by_markets.json: 2048 x 8 136 M, 4096 x 8 184 M, 8192 x 8 202 M end to end, same hyper-parameters)
            lN, lA, lT, liters = int(os.environ.get("CDA_BENCH_LEAGUE_MARKETS", 8192)), 8, 64, 6
            lenv = CDAVecEnv({"num_of_agents": lA, "init_cash": 1000000, "max_step": lT, "is_render": False, "auto_reset": True}, n_markets=lN, device=str(device), with_info=False)
            _, lg, lh = train_league_fused(lenv, iters=liters, horizon=lT, num_trainable=2, log=lambda s_: None)
            tail = lh[2:]
            league_leg = {"value": sum(h["agent_steps"] for h in tail) / sum(h["rollout_s"] + h["update_s"] for h in tail), "iterations": liters, "timed": len(tail),
                          "rollout_ms": statistics.median(h["rollout_s"] for h in tail) * 1e3, "update_ms": statistics.median(h["update_s"] for h in tail) * 1e3,
                          "champions": len(lg.history), "flagged": int((lenv.flags() != 0).sum().item()), "shape": (lN, lA, lT)}
            lenv.close()
        except Exception as ex:  # noqa: BLE001
            league_leg = {"error": repr(ex)}

    if rank == 0:
        import types
        out = headline_line(args, types.SimpleNamespace(world=world, n_gpus=n_gpus, N=N, A=A, K=K, W=W, R=R, took=took, kms=kms, head_ranges=head_ranges, head_groups=head_groups, headline_info=headline_info, gather=gather, tile=tile, spill=spill, spill_wanted=spill_wanted, peak_orders=peak_orders, primer_steps=primer_steps, head_transport=head_transport, head_transport_note=head_transport_note, n_flagged=n_flagged, total_steps=total_steps, MAX_RESIDENT=MAX_RESIDENT, policy_leg=policy_leg, league_leg=league_leg, rr_leg=rr_leg, extras=extras))
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(N, A, args.cpu_seconds, max_step, first_market)
            except Exception as ex:  # noqa: BLE001 - the baseline is a reported extra, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
