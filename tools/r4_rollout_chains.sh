#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD; O=gpurun_out/r4s; mkdir -p $O
for c in 1 2 4 1 2 4; do
  timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --chains $c --out $O/ppo_c$c.json > $O/ppo_c$c.log 2>&1
  python - <<PY
import json
p=json.load(open("$O/ppo_c$c.json"))
it=p["iterations"][1:]
print("chains $c: e2e %.1f M  rollout %.3f ms  update %.3f ms" % (p["value"]/1e6, sum(h["rollout_s"] for h in it)/len(it)*1e3, sum(h["update_s"] for h in it)/len(it)*1e3))
PY
done
