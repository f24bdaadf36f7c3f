#!/usr/bin/env python3
"""Soak of the fused rollouts against the CPU oracle: random configurations (markets, agents, history depth, episode length, horizon, chains, graphs or direct
launches, one shared policy or a league with trainable / frozen / random slots, episode-end capture, stored distributions); every rollout's recorded actions are
replayed through the oracle and must reproduce the recorded observations (f32 bits), rewards (f64 bits) and flags of EVERY step, resets included, and the captured
last observations of the episodes that ended.  The oracle is the checker here, as in tests/ (this tool is test infrastructure).

    python tools/rollout_soak.py --configs 40 --seed 1 > profiles/r05/rollout_soak.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

ACTION_KEYS = ("category", "size_mean", "size_sigma", "price", "price_offset")
DEV = "cuda:0"


def replay(cfg, N, seed, b, T):
    import oracle_lib as O
    ora = O.OracleEnv({key: v for key, v in cfg.items() if key != "auto_reset"}, N)
    o0 = ora.reset(seeds=(seed + np.arange(N)).astype(np.uint64))
    got0 = b["obs"][0].numpy()
    if not np.array_equal(got0.view(np.uint32), o0.view(np.uint32)):
        bad = np.nonzero((got0.view(np.uint32) != o0.view(np.uint32)).any(1))[0]
        j = int(bad[0]); c = np.nonzero(got0[j].view(np.uint32) != o0[j].view(np.uint32))[0]
        raise AssertionError(f"first observation: {len(bad)} of {N} markets differ (first: {bad[:8]}); market {j} columns {c[:8]}: got {got0[j][c[:8]]} want {o0[j][c[:8]]}; "
                             f"nonzero got {int((got0 != 0).sum())} want {int((o0 != 0).sum())}")
    finals, steps = {}, 0
    for t in range(T):
        oo, orw, ot, otr, _ = ora.step(*[b[key][t].numpy() for key in ACTION_KEYS])
        assert np.array_equal(b["reward"][t].numpy().view(np.uint64), orw.view(np.uint64)), ("reward", t)
        assert np.array_equal(b["terminated"][t].numpy(), ot) and np.array_equal(b["truncated"][t].numpy(), otr), ("flags", t)
        done = (ot | otr).astype(bool)
        for j in np.nonzero(done)[0]:
            finals[(t, int(j))] = oo[j].copy()
        if done.any():
            oo = ora.reset(mask=done.astype(np.uint8)).copy()
        assert np.array_equal(b["obs"][t + 1].numpy().view(np.uint32), oo.view(np.uint32)), ("obs", t)
        steps += N
    ora.close()
    return finals, steps


QUIET = False


def say(*a, **k):
    if not QUIET:
        print(*a, **k)


def one(rng, index):
    from gym_continuousdoubleauction_amd import CDAVecEnv, mlp
    H = int(rng.choice([1, 2, 3, 4, 6, 7, 8]))
    A = int(rng.choice([2, 3, 4, 5, 8, 12, 16]))
    N = int(rng.choice([33, 64, 96, 160, 257, 1024]))
    T = int(rng.integers(5, 25))
    max_step = int(rng.choice([T // 2 + 1, T, 3 * T, 4096]))
    league = bool(rng.integers(0, 2))
    groups = int(rng.integers(1, 5))
    graphs = bool(rng.integers(0, 2))
    with_dist, capture = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    cash = int(rng.choice([1000000, 20000, 3000]))                      # small accounts: bankruptcies (terminations) inside the rollout
    seed = int(rng.integers(1, 1 << 30))
    sd = bool(rng.integers(0, 2))                                       # the state-dependent log-std head on every network of the rollout
    tick = int(rng.choice([1, 1, 5]))
    cfg = {"num_of_agents": A, "init_cash": cash, "max_step": max_step, "is_render": False, "auto_reset": True, "n_hist": H, "tick_size": tick}
    say(f"  {index:3d}: {N:3d} markets x {A:2d} agents, n_hist {H}, max_step {max_step:4d}, init_cash {cash:7d}, horizon {T:2d}, {groups} chain(s), "
          f"{'graphs' if graphs else 'direct'}, {'league' if league else 'one shared policy'}, dist {int(with_dist)}, capture {int(capture)}, log-std head {int(sd)}, tick {tick}, seed {seed}", flush=True)
    env = CDAVecEnv(cfg, n_markets=N, with_info=False)
    gen = torch.Generator().manual_seed(seed)
    if league:
        k = int(rng.integers(1, min(A, 3) + 1))
        F = int(rng.integers(0, 4))
        actor = mlp.PolicyBank(DEV, N, A, k, max_frozen=max(F, 1), seed=seed % 1000, random_seed=seed % 7777, n_hist=H)
        for p in range(k):
            th = mlp.init_theta(42 * H, generator=gen, state_dependent_log_std=sd); th[:actor.policies[p].L.OFF_LS] *= 1.5
            actor.policies[p].theta.copy_(th); actor.policies[p].pack()
        for f in range(F):
            row = actor.snapshot(0)
            th = mlp.init_theta(42 * H, generator=gen, state_dependent_log_std=sd); th[:actor.policies[0].L.OFF_LS] *= 1.5
            actor.theta[row].copy_(th)
            actor.wb[row].copy_(mlp.FusedPolicy(DEV, theta=th, n_hist=H).wb)
        sn = torch.randint(-1, k + F, (N, A), generator=gen, dtype=torch.int32)
        sn[:, :k] = torch.arange(k, dtype=torch.int32)
        actor.set_slots(sn)
        what = f"league k={k} frozen={F}"
    else:
        th = mlp.init_theta(42 * H, generator=gen, state_dependent_log_std=sd); th[:mlp.layout(H).OFF_LS] *= 1.5
        actor = mlp.FusedPolicy(DEV, theta=th, n_hist=H)
        what = "one shared policy"
    env.reset(seed=seed)
    roll = mlp.RolloutChains(env, actor, T, groups=groups, seed=seed ^ 0x5555, use_graphs=graphs, capture_ends=capture, with_dist=with_dist)
    steps = ended = 0
    seeds_now = seed
    for rnd in range(2):
        if rnd == 1:                                                   # a second rollout continues the episodes: replayed from a fresh reset of both sides
            env.reset(seed=seed + 1000003); seeds_now = seed + 1000003
        buf = roll.run()
        torch.cuda.synchronize()
        b = {key: v.cpu() for key, v in buf.items() if torch.is_tensor(v)}
        finals, n = replay(cfg, N, seeds_now, b, T)
        steps += n; ended += len(finals)
        if capture:
            fi = b["fin_index"].numpy()
            done = (b["terminated"] | b["truncated"]).numpy().astype(bool)
            assert np.array_equal(fi >= 0, done) and int(b["fin_count"]) == int(done.sum()) == len(finals)
            for (t, j), want in finals.items():
                assert np.array_equal(b["fin_obs"][fi[t, j]].numpy().view(np.uint32), want.view(np.uint32)), ("captured observation", t, j)
    assert (env.flags() == 0).all() and (env.check_invariants() == 0).all()
    env.close()
    say(f"       {what}: {steps} market-steps, {ended} episode ends - ok", flush=True)
    return steps, ended, (H, A, league, graphs, capture, with_dist)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--quiet", action="store_true", help="only the summary (and the configuration that fails)")
    a = ap.parse_args()
    global QUIET
    QUIET = a.quiet
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    steps = ended = 0
    from collections import Counter
    cover = Counter()
    print(f"fused rollouts against the CPU oracle, {a.configs} random configurations (seed {a.seed}):")
    for i in range(a.configs):
        state = rng.bit_generator.state
        try:
            s, e, (H, A, league, graphs, capture, dist) = one(rng, i)
        except BaseException:
            print(f"configuration {i} FAILED (generator state before it: {state})", flush=True)
            raise
        steps += s; ended += e
        cover[f"n_hist {H}"] += 1; cover[f"{A} agents"] += 1; cover["league" if league else "one shared policy"] += 1
        cover["graphs" if graphs else "direct launches"] += 1; cover["episode-end capture"] += int(capture); cover["stored distributions"] += int(dist)
    print("configurations per dimension: " + ", ".join(f"{k}: {v}" for k, v in sorted(cover.items())))
    print(f"{a.configs} configurations, {steps} market-steps, {ended} episode ends replayed bit for bit in {time.time() - t0:.0f} s: no difference")


if __name__ == "__main__":
    main()
