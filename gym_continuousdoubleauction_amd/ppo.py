"""A small PyTorch-ROCm PPO loop on the batched env (BASELINE config #5).

The reference trains through RLlib's PPO (train/train.py:453-541, config/train_config.json `ppo` group:
256x256 tanh MLP, lr 5e-5, 4 epochs, separate value network).  This is NOT a port of that harness - it is
the consumer-side counterpart of the vectorised env: rollouts never leave the GPU (obs/reward tensors come
straight from `CDAVecEnv.step`), one shared policy plays every agent slot (self-play), and the Dict action
is produced by three categorical heads (category 9, price 10, price_offset 3) and two bounded Gaussian heads
(size_mean in [-1,1], size_sigma in [0,1]).

    python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 4
"""
import argparse
import json
import math
import time

import torch
import torch.nn as nn

CAT_N, PRICE_N, OFF_N = 9, 10, 3


class ActorCritic(nn.Module):
    """Separate policy and value MLPs (256x256 tanh), as in config/train_config.json:49."""

    def __init__(self, obs_dim, hidden=256):
        super().__init__()
        def mlp(out):
            return nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, out))
        self.pi = mlp(CAT_N + PRICE_N + OFF_N + 2)
        self.v = mlp(1)
        self.log_std = nn.Parameter(torch.full((2,), -0.5))

    def dists(self, obs):
        o = self.pi(obs)
        # validate_args=False: the argument checks read a flag back to the host, which neither a captured HIP graph
        # nor an asynchronous rollout can afford
        cat = torch.distributions.Categorical(logits=o[:, :CAT_N], validate_args=False)
        price = torch.distributions.Categorical(logits=o[:, CAT_N:CAT_N + PRICE_N], validate_args=False)
        off = torch.distributions.Categorical(logits=o[:, CAT_N + PRICE_N:CAT_N + PRICE_N + OFF_N], validate_args=False)
        mu = o[:, -2:]
        cont = torch.distributions.Normal(mu, self.log_std.exp().expand_as(mu), validate_args=False)
        return cat, price, off, cont

    def act(self, obs):
        cat, price, off, cont = self.dists(obs)
        # Normal.sample() checks std >= 0 on the host (a sync, illegal inside a captured graph): draw the noise directly
        a_cat, a_price, a_off = cat.sample(), price.sample(), off.sample()
        a_cont = (cont.loc + cont.scale * torch.randn_like(cont.loc)).detach()
        logp = cat.log_prob(a_cat) + price.log_prob(a_price) + off.log_prob(a_off) + cont.log_prob(a_cont).sum(-1)
        return (a_cat, a_price, a_off, a_cont), logp, self.v(obs).squeeze(-1)

    def evaluate(self, obs, actions):
        a_cat, a_price, a_off, a_cont = actions
        cat, price, off, cont = self.dists(obs)
        logp = cat.log_prob(a_cat) + price.log_prob(a_price) + off.log_prob(a_off) + cont.log_prob(a_cont).sum(-1)
        ent = cat.entropy() + price.entropy() + off.entropy() + cont.entropy().sum(-1)
        return logp, ent, self.v(obs).squeeze(-1)


def to_env_actions(actions, n, a):
    """Policy sample -> the env's five [N,A] tensors (raw Gaussian heads squashed into the Box bounds)."""
    a_cat, a_price, a_off, a_cont = actions
    mean = torch.tanh(a_cont[:, 0])
    sigma = torch.sigmoid(a_cont[:, 1])
    return (a_cat.view(n, a).to(torch.int32), mean.view(n, a).float(), sigma.view(n, a).float(),
            a_price.view(n, a).to(torch.int32), a_off.view(n, a).to(torch.int32))


def gae(rew, val, last_val, done, gamma=0.99, lam=0.95):
    """rew/val/done: [T, B]; returns advantages and returns [T, B]."""
    T = rew.shape[0]
    adv = torch.zeros_like(rew)
    nxt, run = last_val, torch.zeros_like(last_val)
    for t in range(T - 1, -1, -1):
        nd = 1.0 - done[t]
        delta = rew[t] + gamma * nxt * nd - val[t]
        run = delta + gamma * lam * nd * run
        adv[t] = run
        nxt = val[t]
    return adv, adv + val


def ppo_update(model, opt, obs, actions, logp_old, adv, ret, epochs=4, minibatch=65536, clip=0.2, vf_coef=0.5, ent_coef=0.01, amp=False):
    """amp: run the two MLPs' matrix products of the update in bfloat16 on the MFMA units (torch.autocast; softmax / log-prob /
    losses stay float32, parameters and Adam state stay float32) - the update is the learner-bound 80 % of an iteration."""
    B = obs.shape[0]
    adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    stats = {}
    for _ in range(epochs):
        perm = torch.randperm(B, device=obs.device)
        for s in range(0, B, minibatch):
            idx = perm[s:s + minibatch]
            with torch.autocast(obs.device.type, dtype=torch.bfloat16, enabled=amp):
                logp, ent, v = model.evaluate(obs[idx], tuple(x[idx] for x in actions))
            logp, ent, v = logp.float(), ent.float(), v.float()
            ratio = (logp - logp_old[idx]).exp()
            pg = -torch.min(ratio * adv[idx], ratio.clamp(1 - clip, 1 + clip) * adv[idx]).mean()
            vl = (v - ret[idx]).pow(2).mean()
            loss = pg + vf_coef * vl - ent_coef * ent.mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            opt.step()
            stats = {"pg_loss": pg.detach(), "v_loss": vl.detach(), "entropy": ent.mean().detach()}
    return {k: float(v) for k, v in stats.items()}          # one host sync per update, not one per minibatch


def _join(env):
    """A groups > 1 env leaves each group's outputs on that group's stream: this loop consumes them on the current one."""
    if getattr(env, "groups", 1) > 1:
        env.join()


def _capture_policy_step(model, env, N, A):
    """HIP graph of: observation broadcast -> policy/value forward -> sampling -> env action tensors."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3):                                           # warm-up outside capture (allocator, lazy init)
            pobs = env.obs.repeat_interleave(A, dim=0)
            actions, logp, val = model.act(pobs)
            to_env_actions(actions, N, A)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        pobs = env.obs.repeat_interleave(A, dim=0)
        actions, logp, val = model.act(pobs)
        env_acts = tuple(x.contiguous() for x in to_env_actions(actions, N, A))
    return g, (pobs, actions, logp, val, env_acts)


def train(env, iters=4, horizon=64, lr=5e-5, epochs=4, reward_scale=1e-3, seed=0, log=print, use_graph=True, rollout_hook=None, amp=None):
    """On-device PPO over a CDAVecEnv-shaped env. Returns per-iteration stats (incl. agent-steps/s).
    rollout_hook(iteration, step, env_actions, obs, reward, terminated, truncated): called after every env step with the
    five [N,A] action tensors the policy produced and the step's output tensors (device tensors; clone what you keep)."""
    torch.manual_seed(seed)
    dev = env.obs.device
    amp = (dev.type == "cuda") if amp is None else bool(amp)
    N, A = env.n_markets, env.num_agents
    model = ActorCritic(env.obs_dim).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=lr, fused=dev.type == "cuda")     # one kernel per step instead of one per tensor
    env.reset(seed=seed)
    auto_reset = bool(getattr(env, "config", {}).get("auto_reset", False))
    history = []
    # The policy step of the rollout (MLP forward, five samplers, log-probabilities: ~60 small kernels) is launch bound
    # next to a 55-us env step, so it is captured ONCE in a HIP graph that reads the env's own observation buffer and
    # replayed every step; the env step itself is enqueued between replays on the same stream.
    policy_step = None
    if dev.type == "cuda" and use_graph:
        try:
            policy_step = _capture_policy_step(model, env, N, A)
        except Exception as e:  # noqa: BLE001 - eager rollouts are always available
            log(json.dumps({"hip_graph": f"capture failed, eager rollout: {e}"}))
    for it in range(iters):
        t0 = time.perf_counter()
        buf_obs, buf_act, buf_logp, buf_val, buf_rew, buf_done = [], [], [], [], [], []
        for _ in range(horizon):
            if policy_step is not None:
                g, (pobs_s, actions_s, logp_s, val_s, env_acts_s) = policy_step
                g.replay()                                           # reads env.obs (this step's observation)
                pobs, actions, logp, val = pobs_s.clone(), tuple(x.clone() for x in actions_s), logp_s.clone(), val_s.clone()
                o, r, term, trunc, _ = env.step(*env_acts_s)
                _join(env)
                if rollout_hook is not None:
                    rollout_hook(it, len(buf_obs), env_acts_s, o, r, term, trunc)
            else:
                pobs = env.obs.repeat_interleave(A, dim=0)           # every agent of a market sees the same vector
                with torch.no_grad():
                    actions, logp, val = model.act(pobs)
                env_acts = to_env_actions(actions, N, A)
                o, r, term, trunc, _ = env.step(*env_acts)
                _join(env)
                if rollout_hook is not None:
                    rollout_hook(it, len(buf_obs), env_acts, o, r, term, trunc)
            done = (term | trunc)
            buf_obs.append(pobs); buf_act.append(actions); buf_logp.append(logp); buf_val.append(val)
            buf_rew.append((r.float() * reward_scale).reshape(-1)); buf_done.append(done.repeat_interleave(A).float())
            if not auto_reset and bool(done.any()):                # (a host sync per step; auto_reset envs reset on the device)
                env.reset(mask=done)                               # seed=None semantics: streams continue
        obs = env.obs
        with torch.no_grad():
            last_val = model.v(obs.repeat_interleave(A, dim=0)).squeeze(-1)
        rew, val, dn = torch.stack(buf_rew), torch.stack(buf_val), torch.stack(buf_done)
        adv, ret = gae(rew, val, last_val, dn)
        flat = lambda xs: torch.cat(xs, 0)                         # noqa: E731
        acts = tuple(flat([b[i] for b in buf_act]) for i in range(4))
        t_roll = time.perf_counter()
        stats = ppo_update(model, opt, flat(buf_obs), acts, flat(buf_logp), adv.reshape(-1), ret.reshape(-1), epochs=epochs, amp=amp)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        stats.update(iter=it, mean_reward=float(rew.mean()) / reward_scale, agent_steps=N * A * horizon,
                     agent_steps_per_s=N * A * horizon / (t1 - t0), rollout_s=t_roll - t0, update_s=t1 - t_roll)
        history.append(stats)
        log(json.dumps(stats))
    return model, history


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--markets", type=int, default=4096)
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--horizon", type=int, default=64)
    p.add_argument("--iters", type=int, default=4)
    p.add_argument("--max-step", type=int, default=4096)
    p.add_argument("--fp32-update", action="store_true", help="PPO update in float32 instead of bfloat16 autocast")
    p.add_argument("--out", default=None, help="write a JSON summary (config, per-iteration stats, end-of-run env checks) to this file")
    args = p.parse_args(argv)
    from .vec_env import CDAVecEnv
    env = CDAVecEnv({"num_of_agents": args.agents, "init_cash": 1000000, "max_step": args.max_step, "is_render": False, "auto_reset": True},
                    n_markets=args.markets, device="cuda:0", with_info=False)
    _, hist = train(env, iters=args.iters, horizon=args.horizon, amp=not args.fp32_update)
    flags = env.flags()
    _, bad = env.nav_conservation()
    summary = {"metric": "agent-steps/sec end to end (rollout + PPO update), BASELINE configs[4]",
               "config": {"workload": f"{args.markets} markets x {args.agents} agents, PyTorch-ROCm PPO policy in the loop (256x256 tanh actor and critic, "
                                      f"4 epochs, 65536-sample minibatches), horizon {args.horizon}, {args.iters} iterations, auto_reset on",
                          "markets": args.markets, "agents": args.agents, "horizon": args.horizon, "iters": args.iters, "update_dtype": "float32" if args.fp32_update else "bfloat16 autocast (float32 parameters, Adam state, softmax and losses)"},
               "iterations": hist,
               "value": sum(h["agent_steps"] for h in hist[1:] or hist) / sum(h["rollout_s"] + h["update_s"] for h in hist[1:] or hist),
               "rollout_agent_steps_per_s": sum(h["agent_steps"] for h in hist[1:] or hist) / sum(h["rollout_s"] for h in hist[1:] or hist),
               "unit": "agent-steps/s", "flagged_markets": int((flags != 0).sum().item()), "nav_conservation_violations": int(bad.sum().item()),
               "invariant_violations": int((env.check_invariants() != 0).sum().item())}
    print(json.dumps(summary))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(summary, fh, indent=1)
    env.close()


if __name__ == "__main__":
    main()
