"""League self-play on the batched env (SURVEY §8(f)-1 + 3 together): the trainable slots learn with PPO against a pool of
fixed random opponents and frozen champion snapshots, assigned per episode by the reference's mapping rule.

What the reference does with RLlib (train/train.py:453-541 + train/callbk/league_based_self_play_callback.py: opponent
sampling :1286-1344, champion snapshots :938-1170) and what this module keeps of it:
  * slots below `num_trainable` are played by the learning policy, every other slot by a module drawn from the pool
    (`LeagueSlotMapper`, weights original_opponent_weight / champion_weight, keyed by the episode id);
  * only the trainable slots' transitions feed the PPO update;
  * a champion is a frozen copy of the learning policy, added to the pool when the iteration's mean return of the
    trainable slots beats the best so far by `promote_margin` (a plain statement of the idea, not the reference's full
    trigger logic with its win-rate windows and checkpoint handling).
One iteration = one whole episode of every market (`horizon == max_step`), so all markets change opponents together and the
rollout needs no host sync.

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


def main(argv=None):
    p = argparse.ArgumentParser(description="League self-play (PPO vs random opponents and champion snapshots) on one MI355X.")
    p.add_argument("--markets", type=int, default=1024)
    p.add_argument("--agents", type=int, default=4)
    p.add_argument("--episode", type=int, default=64, help="episode length = max_step")
    p.add_argument("--iters", type=int, default=6)
    args = p.parse_args(argv)
    from .vec_env import CDAVecEnv
    env = CDAVecEnv({"num_of_agents": args.agents, "init_cash": 1000000, "max_step": args.episode, "is_render": False},
                    n_markets=args.markets, device="cuda:0", with_info=False)
    train_league(env, iters=args.iters)
    env.close()


if __name__ == "__main__":
    main()
