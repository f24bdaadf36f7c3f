"""League self-play on the batched env (SURVEY §8(f)-1 + 3 together): the trainable slots learn with PPO against a pool of
fixed random opponents and frozen champion snapshots, assigned per episode by the reference's mapping rule.

Two loops:
  * `train_league_fused` - the reference's training topology (train/train.py:466-503: `num_trained_agents` SEPARATELY trained policies, policy_p plays slot p; every
    other slot drawn per episode from uniform random modules and champions, callbk/league_based_self_play_callback.py:1286-1344; promotion by the reference's
    rule, :780-880) on the hand-written network kernels (include/cda_mlp.h `cda_league`): one policy launch per step for every module of every market, rollouts
    replayed from HIP graphs, each policy's update reading its own slot's records in place.  This is the one `bench.py` and `profiles/r05/bench_league.json` measure.
  * `train_league` - the same idea on the float32 torch network (ppo.ActorCritic), one shared learning policy, step by step from the host: the statement the
    fused loop's tests compare against, 30-50 x slower.  What it keeps of the reference:
      - slots below `num_trainable` are played by the learning policy, every other slot by a module drawn from the pool
        (`LeagueSlotMapper`, weights original_opponent_weight / champion_weight, keyed by the episode id);
      - only the trainable slots' transitions feed the PPO update;
      - a champion is a frozen copy of the learning policy, added to the pool when the iteration's mean return of the trainable slots beats the
        best so far by `promote_margin` (a plain statement of the idea; the reference's trigger is `League.maybe_promote` below);
      - one iteration = one whole episode of every market (`horizon == max_step`).

    python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --episode 64 --iters 10
    python -m gym_continuousdoubleauction_amd.league_train --markets 1024 --agents 4 --iters 6
"""
import argparse
import copy
import json
import time

import torch

from . import ppo
from .league import LeagueSlotMapper, RandomModule


class _FrozenPolicy:
    """A champion snapshot: acts like the policy it was copied from, without gradients."""

    def __init__(self, model):
        self.model = copy.deepcopy(model).eval()
        for p in self.model.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def __call__(self, obs):
        actions, _, _ = self.model.act(obs)
        return tuple(x.reshape(-1) for x in ppo.to_env_actions(actions, obs.shape[0], 1))


def train_league(env, iters=4, num_trainable=1, lr=5e-5, epochs=4, reward_scale=1e-3, seed=0, original_opponent_weight=1.0,
                 champion_weight=3.0, promote_margin=0.0, max_champions=8, recorder=None, log=print):
    """env: CDAVecEnv-shaped, its max_step is the episode length.  Returns (model, mapper, history).
    recorder: an `episode_record.BatchedEpisodeRecorder` (env built with_info=True): the episodes of its markets are written
    in the reference's Parquet schema, `module_id` = the module that played each slot."""
    torch.manual_seed(seed)
    dev = env.obs.device
    N, A, T = env.n_markets, env.num_agents, int(env.max_step)
    k = int(num_trainable)
    model = ppo.ActorCritic(env.obs_dim).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=lr, fused=dev.type == "cuda")
    mapper = LeagueSlotMapper(A, k, A - k, original_opponent_weight, champion_weight)
    modules = {name: RandomModule(dev, seed=1000 + i) for i, name in enumerate(mapper.available_modules[k:])}
    best, history = None, []
    for it in range(iters):
        t0 = time.perf_counter()
        env.reset(seed=seed + it * N)                                   # a fresh episode for every market
        assignment = mapper.assign([f"iter{it}-market{i}" for i in range(N)])
        groups = {name: (torch.as_tensor(mk, device=dev), torch.as_tensor(sl, device=dev))
                  for name, (mk, sl) in mapper.group_by_module(assignment).items() if name in modules}
        if recorder is not None:
            recorder.iteration = it
            names = mapper.names(assignment)
            recorder.begin_episodes([f"iter{it}-market{int(i)}" for i in recorder.markets], module_ids=[list(names[int(i)]) for i in recorder.markets])
        buf_obs, buf_act, buf_logp, buf_val, buf_rew = [], [], [], [], []
        acts = [torch.zeros((N, A), dtype=dt, device=dev) for dt in (torch.int32, torch.float32, torch.float32, torch.int32, torch.int32)]
        for _ in range(T):
            obs = env.obs
            pobs = obs.repeat_interleave(k, dim=0)                       # the k trainable slots of each market
            with torch.no_grad():
                a_tr, logp, val = model.act(pobs)
            for dst, src in zip(acts, ppo.to_env_actions(a_tr, N, k)):
                dst[:, :k] = src
            for name, (mk, sl) in groups.items():                        # one batched forward per opponent module
                for dst, src in zip(acts, modules[name](obs.index_select(0, mk))):
                    dst[mk, sl] = src.to(dst.dtype)
            o_next, r, _, _, info = env.step(*acts)
            if recorder is not None:
                recorder.record_step(o_next, r, info, acts)
            buf_obs.append(pobs); buf_act.append(a_tr); buf_logp.append(logp); buf_val.append(val)
            buf_rew.append((r[:, :k].float() * reward_scale).reshape(-1))
        if recorder is not None:
            recorder.finish(complete=True)
        rew, val = torch.stack(buf_rew), torch.stack(buf_val)
        done = torch.zeros_like(rew)
        done[-1] = 1.0                                                   # the episode ends with the rollout
        adv, ret = ppo.gae(rew, val, torch.zeros_like(val[0]), done)
        flat = lambda xs: torch.cat(xs, 0)                               # noqa: E731
        a_all = tuple(flat([b[i] for b in buf_act]) for i in range(4))
        stats = ppo.ppo_update(model, opt, flat(buf_obs), a_all, flat(buf_logp), adv.reshape(-1), ret.reshape(-1), epochs=epochs,
                               minibatch=min(65536, N * k * T))
        episode_return = float(rew.sum(0).mean()) / reward_scale
        promoted = None
        if (best is None or episode_return > best + promote_margin) and sum(n.startswith("champion_") for n in mapper.available_modules) < max_champions:
            best = episode_return if best is None else max(best, episode_return)
            promoted = mapper.add_champion()
            modules[promoted] = _FrozenPolicy(model)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        stats.update(iter=it, episode_return=episode_return, promoted=promoted, pool=list(mapper.pool()),
                     agent_steps_per_s=N * A * T / (time.perf_counter() - t0))
        history.append(stats)
        log(json.dumps(stats))
    return model, mapper, history


class League:
    """The league's host state next to the device banks: who is in the pool, which bank row holds which champion, when the last one was promoted - the reference's
    trigger (train/callbk/league_based_self_play_callback.py:780-880): a trainable policy becomes a champion when its mean episode return of the iteration exceeds
    mean + std_dev_multiplier x std of all modules' returns, at most every `min_iterations_between_champions` iterations, the oldest champion leaving once
    `max_champions` are held (config/train_config.json:60-64)."""

    def __init__(self, mapper, bank, std_dev_multiplier=0.1, max_champions=8, min_iterations_between_champions=2):
        self.mapper, self.bank = mapper, bank
        self.std_dev_multiplier, self.max_champions, self.min_gap = float(std_dev_multiplier), int(max_champions), int(min_iterations_between_champions)
        self.net_of = {}                                       # champion id -> bank row
        self.history = []                                      # [{"id", "iteration", "return", "source"}]

    def module_returns(self, per_slot, slot_pool, allreduce=None):
        """{module id: mean return of the episodes it completed this iteration}: per_slot f64 [N, A, 2] (mlp.EpisodeReturns), slot_pool i32 [N, A] (the draw per slot,
        -1 = the slot's own trainable policy).  A handful of tiny device reductions + one small copy to the host."""
        k, names = self.mapper.num_trainable, self.mapper.available_modules
        N, A = slot_pool.shape
        idx = torch.where(slot_pool < 0, torch.arange(A, device=slot_pool.device, dtype=torch.int32).expand(N, A), slot_pool + k).long().reshape(-1)
        # (masked sums - not index_add_: a dozen bins under 16 k double-precision atomics took 1.6 ms per call, a fifth of an iteration; not a one-hot matrix product
        # either: a float64 GEMM with two columns is 1.9 ms in the library)
        mask = idx[None, :] == torch.arange(len(names), device=idx.device)[:, None]                    # [modules, N * A]
        flat = per_slot.reshape(-1, 2)
        zero = torch.zeros((), dtype=torch.float64, device=flat.device)
        sums = torch.stack([torch.where(mask, flat[:, 0][None, :], zero).sum(1), torch.where(mask, flat[:, 1][None, :], zero).sum(1)], dim=1)
        if allreduce is not None:                              # data parallel: the modules' returns over ALL shards (every rank then takes the same decision)
            allreduce(sums)
        host = sums.cpu().numpy()
        return {names[i]: host[i, 0] / host[i, 1] for i in range(len(names)) if host[i, 1] > 0}

    def maybe_promote(self, returns, iteration):
        """the reference's end-of-iteration decision; returns the new champion's id or None"""
        import numpy as np
        k = self.mapper.num_trainable
        valid = [v for v in returns.values() if np.isfinite(v)]
        if not valid:
            return None
        threshold = float(np.mean(valid)) + self.std_dev_multiplier * float(np.std(valid))
        best, best_ret = None, -float("inf")
        for p in range(k):
            r = returns.get(self.mapper.available_modules[p])
            if r is not None and r > best_ret:
                best, best_ret = p, r
        if best is None or not best_ret > threshold:
            return None
        if self.history and iteration - self.history[-1]["iteration"] < self.min_gap:
            return None
        slot = None
        if len(self.net_of) >= self.max_champions:            # rolling window: the oldest champion leaves and its bank row is reused
            oldest = next(c["id"] for c in self.history if c["id"] in self.net_of)
            slot = self.net_of.pop(oldest) - k
            self.mapper.remove(oldest)
        cid = self.mapper.add_champion()
        self.net_of[cid] = self.bank.snapshot(best, frozen_slot=slot)
        self.history.append({"id": cid, "iteration": iteration, "return": best_ret, "source": self.mapper.available_modules[best]})
        return cid


def train_league_fused(env, iters=4, horizon=None, num_trainable=2, lr=5e-5, epochs=4, seed=0, original_opponent_weight=1.0, champion_weight=3.0,
                       std_dev_multiplier=0.1, max_champions=8, min_iterations_between_champions=2, chains=4, minibatch=262144, objective=None, use_graph=True,
                       recorder=None, info_markets=0, run_id="league", log=print, keep=None, allreduce=None, world=1, first_market=0, episode_metrics=True,
                       strict_nav_check=True, state_dependent_log_std=False, hidden=(256, 256)):
    """League self-play on the fused kernels (include/cda_mlp.h `cda_league`): the reference's training topology - `num_trainable` SEPARATELY trained policies
    (policy_p plays slot p), every other slot drawn per episode from the pool of uniform random modules and frozen champions by the reference's mapping rule
    (computed on the device, league.LeagueSlotMapper.assign_device) - at the speed of the fused loop: ONE policy launch per step serves every module of every
    market, the rollout never leaves its HIP graphs, each trainable policy's update reads its own slot's sample records in place (record stride, no compaction).
    env: CDAVecEnv with auto_reset; an episode = env.max_step steps = max_step / horizon iterations (horizon must divide it): all markets change opponents
    together at the episode boundary.  Returns (bank, league, history).
    episode_metrics (default on): every episode end is checked and tallied on the device (ppo.train_fused), here PER MODULE - keyed by the draw of the mapping fn, so
    a champion's or a random module's figures are its own whatever slot it played (the reference's `module_episode_returns_mean` keying, callbk:780-800);
    history[i]["episode_metrics"] = {"episodes", "nav_conservation_violations", ..., "modules": {module id: {...}}}; a violation raises (strict_nav_check).
    allreduce / world / first_market: the data-parallel learner of ppo.train_fused for the league - every rank rolls out and back-propagates its own shard of
    markets (global indices [first_market, first_market + N): env seeds, sampling keys and EPISODE IDS follow them, so the opponents a market meets do not depend
    on the GPU count), the ranks sum each policy's gradient (one all-reduce of 0.9 MB per policy and minibatch step) and advantage sums, and the per-module returns
    behind the promotion rule - so every rank promotes the same champions in lockstep."""
    import numpy as np
    from . import ppo
    from .mlp import EpisodeReturns, FusedUpdate, PolicyBank, RolloutChains
    obj = dict(ppo.PPO_DEFAULTS)
    obj.update(objective or {})
    dev = env.obs.device
    N, A, k = env.n_markets, env.num_agents, int(num_trainable)
    T = int(horizon or env.max_step)
    if int(env.max_step) % T:
        raise ValueError("the horizon must divide the episode length (max_step)")
    per_episode = int(env.max_step) // T
    bank = PolicyBank(dev, N, A, k, max_frozen=max_champions, seed=seed, random_seed=seed + 12345 + 104729 * int(first_market), n_hist=env.n_hist, state_dependent_log_std=state_dependent_log_std, hidden=hidden)
    mapper = LeagueSlotMapper(A, k, A - k, original_opponent_weight, champion_weight)
    league = League(mapper, bank, std_dev_multiplier, max_champions, min_iterations_between_champions)
    env.reset(seed=seed + int(first_market))
    if episode_metrics:
        env.enable_episode_metrics(True)
    module_of, module_names = torch.zeros((N, A), dtype=torch.int32, device=dev), list(mapper.available_modules)
    use_kl = obj["kl_coef"] > 0.0
    roll = RolloutChains(env, bank, T, groups=chains, seed=seed + 7919 * int(first_market), use_graphs=use_graph, with_dist=use_kl,
                         capture_ends=bool(obj["bootstrap_truncation"]), info_markets=info_markets if recorder is not None else 0)
    R = T * N
    rows_mb = max(32, min(R, (max(1, minibatch) // 32) * 32))              # one sample per row: a minibatch of `minibatch` samples is that many rows
    dp = allreduce is not None and world > 1
    upds = [FusedUpdate(bank.policies[p], R, rows_mb, 1, allreduce=allreduce if dp else None, world=world if dp else 1) for p in range(k)]
    # the trainable policies' updates are independent of each other: on their own streams one's kernel tails fill with the other's launches (CDA_LEAGUE_UPDATE_STREAMS=0: one stream)
    import os
    update_streams = list(roll.streams[:k]) if (k > 1 and len(roll.streams) >= 2 and not dp and os.environ.get("CDA_LEAGUE_UPDATE_STREAMS", "1") != "0") else []
    # (data parallel: one stream - the ranks must issue their collectives in ONE order)
    returns = EpisodeReturns(N, A, dev, per_slot=True)
    # Opponents change for ALL markets at the `it % per_episode == 0` boundary (lock-step by construction: a market that terminates early - every agent done - auto-resets
    # and plays its next episode against the SAME draw until the boundary; its early end is credited to that draw).  EpisodeReturns.per_slot holds one rollout's completed
    # episodes, so over a window of several rollouts they are summed here.
    window_per_slot = torch.zeros_like(returns.per_slot) if per_episode > 1 else None
    slot_pool = torch.full((N, A), -1, dtype=torch.int32, device=dev)
    kl_coefs = [float(obj["kl_coef"])] * k
    episode_ids = lambda e: [f"{run_id}-episode{e}-market{int(first_market) + i}" for i in range(N)]     # noqa: E731  (global market index)
    next_crcs = mapper.episode_crcs(episode_ids(0))
    if recorder is not None:
        names = lambda: np.array(mapper.available_modules, dtype=object)             # noqa: E731
        recorder.episode_namer = lambda m, e: f"{run_id}-episode{e}-market{m}"
        recorder.init_cash = int(env.config.get("init_cash", 1000000))
    history = []
    for it in range(iters):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if it % per_episode == 0:                                            # a new episode everywhere: new opponents (one launch; the crcs were computed while the GPU worked)
            mapper.assign_device(bank, crcs=next_crcs, net_of=league.net_of, slot_pool=slot_pool)
            if episode_metrics:                                              # who plays what during the coming episodes: index into the pool AS IT STANDS NOW
                torch.where(slot_pool < 0, torch.arange(A, device=dev, dtype=torch.int32).expand(N, A), slot_pool + k, out=module_of)
                module_names = list(mapper.available_modules)
            if recorder is not None:
                sp = slot_pool[N - roll.info_markets:].cpu().numpy() if roll.info_markets else None
                mods = names()
                recorder.module_namer = lambda m: [mods[a] if sp[m - (N - roll.info_markets), a] < 0 else mods[k + sp[m - (N - roll.info_markets), a]] for a in range(A)]
                if hasattr(recorder, "_ordinal"):
                    recorder._name_episodes()
        buf = roll.run()
        rec, stats, count = roll.gae(gamma=obj["gamma"], lam=obj["lam"], reward_scale=obj["reward_scale"])
        if dp:                                                               # the advantages are standardised over the GLOBAL batch of each policy
            allreduce(stats)
            count = count * world
        torch.cuda.synchronize(dev)
        t_roll = time.perf_counter()
        obs_rows = buf["obs"][:T].view(R, -1)
        outs = []
        cur = torch.cuda.current_stream(dev)
        if update_streams:
            fork = torch.cuda.Event(); fork.record(cur)
        for p in range(k):                                                   # policy p learns from slot p's records only: rec + 8 p floats, a row every 8 A
            upds[p].set_extra(rec_stride=8 * A, kl_coef=kl_coefs[p], vf_clip=obj["vf_clip"], dist_old=buf["dist"][p] if use_kl else None,
                              log_std_old=roll.log_std_old[p] if use_kl else None)
            st = update_streams[p % len(update_streams)] if update_streams else cur
            if update_streams:
                st.wait_event(fork)
            with torch.cuda.stream(st):                                      # (the policies' updates are independent chains of launches)
                outs.append(upds[p].run(obs_rows, epochs=epochs, clip=obj["clip"], vf_coef=obj["vf_coef"], ent_coef=obj["ent_coef"], lr=lr, max_norm=obj["max_norm"],
                                        records=(rec.data_ptr() + 32 * p, stats[p], count)))
            if update_streams:
                ev = torch.cuda.Event(); ev.record(st); cur.wait_event(ev)
        returns.update(buf, T)
        if window_per_slot is not None:                                      # an episode window of several rollouts: an early end in an EARLIER rollout of the window counts too
            window_per_slot.add_(returns.per_slot)
        em = env.collect_episode_metrics(module_of=module_of, n_modules=len(module_names)) if episode_metrics else None     # (before the pool changes)
        if (it + 1) % per_episode == 0:                                      # host work under the GPU's: the next episode's ids
            next_crcs = mapper.episode_crcs(episode_ids((it + 1) // per_episode))
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        stats_h = {"iter": it, "agent_steps": N * A * T, "agent_steps_per_s": N * A * T / (t1 - t0), "rollout_s": t_roll - t0, "update_s": t1 - t_roll}
        for p, o in enumerate(outs):
            stats_h[f"policy_{p}"] = {key: float(v) for key, v in o.items()}
            kl_coefs[p] = ppo.adapt_kl_coef(kl_coefs[p], stats_h[f"policy_{p}"]["kl"], obj["kl_target"])
        promoted = None
        if (it + 1) % per_episode == 0:                                      # episodes just ended: credit the modules, run the reference's promotion rule
            mr = league.module_returns(returns.per_slot if window_per_slot is None else window_per_slot, slot_pool, allreduce=allreduce if dp else None)
            if window_per_slot is not None:
                window_per_slot.zero_()
            promoted = league.maybe_promote(mr, it)
            stats_h["module_returns"] = {m: float(v) for m, v in mr.items()}
        stats_h.update(promoted=promoted, pool=list(mapper.pool()), mean_reward_trainable=float(buf["reward"][:, :, :k].mean()))
        roll.check_capture_overflow()                                       # (outside the timed region; warns)
        if recorder is not None and roll.info is not None:
            recorder.record_rollout(roll, iteration=it)
        if em is not None:
            from . import episode_metrics as EM
            if dp:                                                           # the additive columns over all shards (min / max columns stay this rank's)
                allreduce(em[0]); allreduce(em[1])
            summ = EM.summarise(*em, module_names=module_names)
            stats_h["episode_metrics"] = summ
        history.append(stats_h)
        log(json.dumps(stats_h))
        if em is not None:
            EM.check_nav_conservation(it, summ, strict=strict_nav_check, log=log)
    if keep is not None:
        keep.update(buffers=roll.buf, rollout=roll, updates=upds, slot_pool=slot_pool, returns=returns)
    return bank, league, history


def main(argv=None):
    p = argparse.ArgumentParser(description="League self-play (PPO vs random opponents and champion snapshots) on one MI355X.")
    p.add_argument("--markets", type=int, default=1024)
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--episode", type=int, default=64, help="episode length = max_step")
    p.add_argument("--iters", type=int, default=6)
    p.add_argument("--fused", action="store_true", help="the league on the hand-written network kernels (train_league_fused): the reference's topology at the fused loop's speed")
    p.add_argument("--trainable", type=int, default=None, help="separately trained policies (default: 2 with --fused, the reference's num_trained_agents; 1 otherwise)")
    p.add_argument("--horizon", type=int, default=None, help="--fused: rollout length (divides --episode; default = --episode)")
    p.add_argument("--chains", type=int, default=4)
    p.add_argument("--objective", choices=("ppo", "rllib"), default="ppo")
    p.add_argument("--fcnet-hiddens", type=int, nargs=2, default=(256, 256), metavar=("H1", "H2"), help="hidden widths of the trainable policies (config/train_config.json:49), <= 256 each")
    p.add_argument("--log-std-head", action="store_true", help="the trainable policies carry the state-dependent log-std head (RLlib's default module for Box actions)")
    p.add_argument("--out", default=None, help="write a JSON summary to this file")
    args = p.parse_args(argv)
    from .vec_env import CDAVecEnv
    from . import ppo
    cfg = {"num_of_agents": args.agents, "init_cash": 1000000, "max_step": args.episode, "is_render": False}
    if not args.fused:
        env = CDAVecEnv(cfg, n_markets=args.markets, device="cuda:0", with_info=False)
        train_league(env, iters=args.iters, num_trainable=args.trainable or 1)
        env.close()
        return
    env = CDAVecEnv(dict(cfg, auto_reset=True), n_markets=args.markets, device="cuda:0", with_info=False)
    k = args.trainable or 2
    _, league, hist = train_league_fused(env, iters=args.iters, horizon=args.horizon, num_trainable=k, chains=args.chains,
                                         objective=ppo.RLLIB_DEFAULTS if args.objective == "rllib" else None, state_dependent_log_std=args.log_std_head, hidden=tuple(args.fcnet_hiddens))
    flags = env.flags()
    _, bad = env.nav_conservation()
    tail = hist[2:] if len(hist) >= 4 else (hist[1:] or hist)          # (two warm-up iterations: graph capture, first replays)
    summary = {"metric": "agent-steps/sec end to end (league rollout + one PPO update per trainable policy)",
               "config": {"workload": f"{args.markets} markets x {args.agents} agents, {k} separately trained policies (policy_p plays slot p) against modules drawn per episode "
                                      f"and slot from {args.agents - k} uniform random modules + up to 8 champion snapshots (the reference's mapping rule, on the device); episode "
                                      f"{args.episode} steps, horizon {args.horizon or args.episode}, {args.iters} iterations; hand-written bf16 MFMA network kernels, one policy launch per step "
                                      "for every module, 4 epochs per policy per iteration",
                          "markets": args.markets, "agents": args.agents, "trainable": k, "episode": args.episode, "horizon": args.horizon or args.episode, "chains": args.chains,
                          "objective": args.objective},
               "iterations": hist, "timed_iterations": len(tail),
               "value": sum(h["agent_steps"] for h in tail) / sum(h["rollout_s"] + h["update_s"] for h in tail), "unit": "agent-steps/s",
               "rollout_agent_steps_per_s": sum(h["agent_steps"] for h in tail) / sum(h["rollout_s"] for h in tail),
               "champions": league.history, "flagged_markets": int((flags != 0).sum().item()), "nav_conservation_violations": int(bad.sum().item()),
               "invariant_violations": int((env.check_invariants() != 0).sum().item())}
    print(json.dumps(summary))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(summary, fh, indent=1)
    env.close()


if __name__ == "__main__":
    main()
