#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python -m gym_continuousdoubleauction_amd.ppo --iters 4 2>&1 | grep '"iter"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ppo.main iter', d['iter'], round(d['rollout_s']*1e3,2), round(d['update_s']*1e3,2))"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from gym_continuousdoubleauction_amd import CDAVecEnv, ppo
env = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}, n_markets=4096, with_info=False)
_, h = ppo.train_fused(env, iters=4, horizon=64, log=lambda s: None)
print("direct train_fused", [round(x["rollout_s"] * 1e3, 2) for x in h])
env2 = CDAVecEnv({"num_of_agents": 4, "init_cash": 1000000, "max_step": 4096, "is_render": False, "auto_reset": True}, n_markets=4096, with_info=False)
_, h = ppo.train_fused(env2, iters=4, horizon=64, log=print)
print("direct train_fused, log=print", [round(x["rollout_s"] * 1e3, 2) for x in h])
PY
