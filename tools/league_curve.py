#!/usr/bin/env python3
"""Does the fused LEAGUE loop learn?  The reference's topology (8 agents, 2 separately trained policies, the other slots drawn per episode from uniform random modules and
champion snapshots) with short episodes (episode = horizon: every iteration plays whole episodes from their first step, so the returns of consecutive iterations are
comparable) - mean episode return per MODULE and iteration, and the champions the reference's promotion rule makes on the way.

    python tools/league_curve.py [--markets 1024] [--episode 32] [--iters 80] [--lr 3e-4] > profiles/r05/league_learning_curve.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def curves(markets=512, agents=8, episode=32, iters=40, lr=3e-4, seed=0, objective=None):
    """Mean episode return per iteration of the two trained policies on the fused league loop, and of the legacy float32 torch league loop (league_train.train_league:
    library GEMMs, autograd, torch.optim.Adam; ONE float32 network plays both trainable slots there, champions by its best-so-far rule) on the same env shape, seed,
    learning rate and episode length.  {"policy_0": [...], "policy_1": [...], "legacy": [...], "champions": n, "legacy_champions": n}"""
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd.league_train import train_league, train_league_fused
    cfg = {"num_of_agents": agents, "init_cash": 1000000, "max_step": episode, "is_render": False, "auto_reset": True}
    env = CDAVecEnv(cfg, n_markets=markets, with_info=False)
    _, league, hist = train_league_fused(env, iters=iters, horizon=episode, num_trainable=2, lr=lr, seed=seed, objective=objective, log=lambda s: None)
    clean = bool((env.flags() == 0).all() and (env.check_invariants() == 0).all())
    env.close()
    env = CDAVecEnv(dict(cfg, auto_reset=False), n_markets=markets, with_info=False)
    _, mapper, lh = train_league(env, iters=iters, num_trainable=2, lr=lr, seed=seed, log=lambda s: None)
    clean = clean and bool((env.flags() == 0).all())
    env.close()
    return {"policy_0": [h["module_returns"]["policy_0"] for h in hist], "policy_1": [h["module_returns"]["policy_1"] for h in hist],
            "random": [[v for k, v in h["module_returns"].items() if not k.startswith("champion_") and k not in ("policy_0", "policy_1")] for h in hist],
            "legacy": [h["episode_return"] for h in lh], "champions": len(league.history),
            "legacy_champions": sum(1 for h in lh if h.get("promoted")), "clean": clean}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--markets", type=int, default=1024)
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--episode", type=int, default=32)
    ap.add_argument("--iters", type=int, default=80)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--against-float32", action="store_true", help="the fused loop's two policies beside the legacy float32 torch league loop (curves())")
    ap.add_argument("--rllib", action="store_true", help="with --against-float32: the fused loop optimises ppo.RLLIB_DEFAULTS")
    a = ap.parse_args()
    if a.against_float32:
        from gym_continuousdoubleauction_amd import ppo
        c = curves(a.markets, a.agents, a.episode, a.iters, a.lr, objective=dict(ppo.RLLIB_DEFAULTS) if a.rllib else None)
        print(f"league loop, fused kernels (2 separately trained policies{', RLLIB_DEFAULTS objective' if a.rllib else ''}) against the legacy float32 torch loop (one network on both trainable slots): "
              f"{a.markets} markets x {a.agents} agents, episode = horizon = {a.episode}, lr {a.lr}, {a.iters} iterations")
        print(f"{'iter':>5s} {'policy_0':>12s} {'policy_1':>12s} {'float32 torch':>14s}")
        for i in range(a.iters):
            if i % 5 == 0 or i >= a.iters - 3:
                print(f"{i:5d} {c['policy_0'][i]:12.1f} {c['policy_1'][i]:12.1f} {c['legacy'][i]:14.1f}")
        m = lambda x, sl: sum(x[sl]) / 3          # noqa: E731
        for k in ("policy_0", "policy_1", "legacy"):
            print(f"{k}: first three {m(c[k], slice(0, 3)):.1f} -> last three {m(c[k], slice(-3, None)):.1f}  (recovered {(1 - m(c[k], slice(-3, None)) / m(c[k], slice(0, 3))) * 100:.2f} %)")
        print("champions:", c["champions"], "(fused, the reference's rule)", c["legacy_champions"], "(legacy, best-so-far rule); clean:", c["clean"])
        return
    from gym_continuousdoubleauction_amd import CDAVecEnv
    from gym_continuousdoubleauction_amd.league_train import train_league_fused
    env = CDAVecEnv({"num_of_agents": a.agents, "init_cash": 1000000, "max_step": a.episode, "is_render": False, "auto_reset": True}, n_markets=a.markets, with_info=False)
    _, league, hist = train_league_fused(env, iters=a.iters, horizon=a.episode, num_trainable=2, lr=a.lr, log=lambda s: None)
    print(f"league self-play on the fused kernels: {a.markets} markets x {a.agents} agents, 2 trained policies, episode = horizon = {a.episode} steps, lr {a.lr}, {a.iters} iterations")
    print("mean episode return per module (the trainable policies; the mean over the uniform random modules; the mean over the champions in the pool), champions promoted")
    print(f"{'iter':>5s} {'policy_0':>12s} {'policy_1':>12s} {'random modules':>16s} {'champions':>12s}   pool")
    for h in hist:
        mr = h.get("module_returns") or {}
        rnd = [v for k, v in mr.items() if not k.startswith("champion_") and k not in ("policy_0", "policy_1")]     # (the fixed opponents are the reference's policy_2 .. policy_7: uniform random modules)
        ch = [v for k, v in mr.items() if k.startswith("champion_")]
        if h["iter"] % 5 == 0 or h.get("promoted") or h["iter"] == a.iters - 1:
            f = lambda x: f"{x:12.1f}" if x is not None else f"{'-':>12s}"          # noqa: E731
            print(f"{h['iter']:5d} {f(mr.get('policy_0'))} {f(mr.get('policy_1'))} {(sum(rnd) / len(rnd) if rnd else float('nan')):16.1f} {f(sum(ch) / len(ch) if ch else None)}   "
                  f"{len([m for m in h['pool'] if m.startswith('champion_')])} champion(s)" + (f"  <- promoted {h['promoted']}" if h.get("promoted") else ""))
    first = [hist[i]["module_returns"] for i in range(3)]
    last = [hist[-1 - i]["module_returns"] for i in range(3)]
    for p in ("policy_0", "policy_1"):
        print(f"{p}: first three iterations {sum(m[p] for m in first) / 3:.1f} -> last three {sum(m[p] for m in last) / 3:.1f}")
    print("champions:", [(c["id"], c["iteration"], round(c["return"], 1), c["source"]) for c in league.history])
    print("flagged markets:", int((env.flags() != 0).sum().item()), " invariant violations:", int((env.check_invariants() != 0).sum().item()))
    env.close()


if __name__ == "__main__":
    main()
