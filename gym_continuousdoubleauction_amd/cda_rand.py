"""Random-agent driver for the MI355X env (the role `CDA_rand.run_random` plays in the reference,
gym_continuousDoubleAuction/CDA_rand.py:40-85: reset with a seed, draw every agent's action from its action
space, step until the horizon or an `__all__` flag).

Two modes:
  * `run_random(...)`   one market through the dict-shaped `CDAEnv` facade (BASELINE config #1 plumbing);
  * `run_random_batched(...)`  N markets through `CDAVecEnv` with the same uniform action law sampled on the GPU.

    python -m gym_continuousdoubleauction_amd.cda_rand --agents 4 --steps 1000 --seed 123 [--markets 4096]
"""
import argparse
import sys
import time

# config/cli_defaults.json:9-14 of the reference
DEFAULT_AGENTS, DEFAULT_STEPS, DEFAULT_CASH = 4, 1000, 1000000


def _env_config(num_agents, max_step, init_cash):
    return {"num_of_agents": DEFAULT_AGENTS if num_agents is None else num_agents,
            "max_step": DEFAULT_STEPS if max_step is None else max_step,
            "init_cash": DEFAULT_CASH if init_cash is None else init_cash, "is_render": False}


def run_random(num_agents=None, max_step=None, init_cash=None, is_render=None, seed=None, device="cuda:0"):
    """Single market, dict API.  Returns the number of steps taken (the reference's contract)."""
    from .env import CDAEnv
    cfg = _env_config(num_agents, max_step, init_cash)
    env = CDAEnv(cfg, device=device)
    try:
        env.reset(seed=seed)
        space = env.action_spaces[env.agents[0]]            # one shared Dict space for every agent
        if seed is not None:
            space.seed(seed)
        taken = 0
        while taken < cfg["max_step"]:
            _, _, terminateds, truncateds, _ = env.step({aid: space.sample() for aid in env.agents})
            taken += 1
            if terminateds["__all__"] or truncateds["__all__"]:
                break
        return taken
    finally:
        env.close()


def run_random_batched(n_markets, num_agents=None, max_step=None, init_cash=None, seed=0, device="cuda:0", fused=True):
    """N markets; actions ~ the RandomRLModule law (train/model/model_handler.py:38-53) drawn on the device.
    fused=True: the whole episode of every market in ONE launch (`cda_run_random`: state stays in LDS, no market waits for
    the slowest of the batch); fused=False: one launch per step with torch-generated actions.
    Returns (steps, agent_steps_per_second)."""
    import torch
    from .vec_env import CDAVecEnv
    cfg = _env_config(num_agents, max_step, init_cash)
    env = CDAVecEnv(cfg, n_markets=n_markets, device=device, with_info=False)
    n, a = env.n_markets, env.num_agents
    if fused:
        env.reset(seed=seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, term, trunc, taken = env.run_random(cfg["max_step"], action_seed=seed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if not bool((term | trunc).all()):
            raise RuntimeError("every market must end its episode within max_step")
        total = int(taken.sum().item())
        env.close()
        return int(taken.max().item()), total * a / dt
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    env.reset(seed=seed)
    torch.cuda.synchronize()
    t0, steps = time.perf_counter(), 0
    for _ in range(cfg["max_step"]):
        _, _, term, trunc, _ = env.step(
            torch.randint(0, 9, (n, a), generator=gen, device=device, dtype=torch.int32),
            torch.rand((n, a), generator=gen, device=device) * 2.0 - 1.0,
            torch.rand((n, a), generator=gen, device=device),
            torch.randint(0, 10, (n, a), generator=gen, device=device, dtype=torch.int32),
            torch.randint(0, 3, (n, a), generator=gen, device=device, dtype=torch.int32))
        steps += 1
        if steps == cfg["max_step"] and not bool(trunc.all()):
            raise RuntimeError("every market must truncate exactly on max_step")
    torch.cuda.synchronize()
    rate = n * a * steps / (time.perf_counter() - t0)
    env.close()
    return steps, rate


def main(argv=None):
    ap = argparse.ArgumentParser(description="Random-agent CDA simulation on one MI355X.")
    ap.add_argument("--agents", type=int, default=DEFAULT_AGENTS)
    ap.add_argument("--steps", type=int, default=DEFAULT_STEPS)
    ap.add_argument("--init-cash", type=int, default=DEFAULT_CASH)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--markets", type=int, default=1, help="> 1 runs the batched env")
    ap.add_argument("--per-step-launch", action="store_true", help="batched env: one launch per step instead of one per episode")
    args = ap.parse_args(argv)
    if args.markets > 1:
        steps, rate = run_random_batched(args.markets, args.agents, args.steps, args.init_cash, seed=args.seed or 0,
                                         fused=not args.per_step_launch)
        print(f"{args.markets} markets x {args.agents} random agents: {steps} steps, {rate / 1e6:.1f} M agent-steps/s")
    else:
        steps = run_random(args.agents, args.steps, args.init_cash, seed=args.seed)
        print(f"completed {steps} steps with {args.agents} random agents.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
