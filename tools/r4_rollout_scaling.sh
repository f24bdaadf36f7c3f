#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export PYTHONPATH=$PWD; O=gpurun_out/r4t; mkdir -p $O
for cfg in "1024 1" "2048 1" "2048 2" "4096 1" "8192 1" "8192 4" "16384 4"; do
  set -- $cfg
  timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets $1 --agents 4 --horizon 64 --iters 6 --chains $2 --out $O/ppo.json > $O/ppo.log 2>&1
  python - <<PY
import json
p=json.load(open("$O/ppo.json"))
it=p["iterations"][1:]
r=sum(h["rollout_s"] for h in it)/len(it)
print("markets $1 chains $2: rollout %.3f ms = %.1f us per step; update %.3f ms; e2e %.1f M" % (r*1e3, r*1e6/64, sum(h["update_s"] for h in it)/len(it)*1e3, p["value"]/1e6))
PY
done
