import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
cfg = {"num_of_agents": 4, "init_cash": 1000000, "max_step": 12, "is_render": False, "auto_reset": True}
env2 = CDAVecEnv(cfg, n_markets=64, with_info=False)
env2.reset(seed=5)
m = ppo.ActorCritic(env2.obs_dim).to(env2.device)
orig = ppo.new_sampler_state
keep = {}
def spy(seed, dev):
    st = orig(seed, dev); keep["st"] = st; return st
ppo.new_sampler_state = spy
g, buf, t_dev = ppo._capture_rollout_step(m, env2, 64, 4, 6, seed=9, shared=True)
ppo.new_sampler_state = orig
torch.cuda.synchronize()
print("after capture: counter", keep["st"][1].item(), "t_dev", t_dev.item())
env2.reset(seed=5)
obs0 = env2.obs.clone()
g.replay(); torch.cuda.synchronize()
print("after replay: counter", keep["st"][1].item(), "t_dev", t_dev.item(), "obs slot equal", torch.equal(buf["obs"][0], obs0))
for c in range(0, 8):
    st = ppo.new_sampler_state(9, env2.device); st[1].fill_(c)
    with torch.no_grad():
        acts, logp, val, env_acts = m.act_fused(obs0, 64, 4, st, shared=True)
    torch.cuda.synchronize()
    print(c, torch.equal(buf["a_cat"][0], acts[0]), float((buf["logp"][0] - logp).abs().max()), "counter now", st[1].item())
