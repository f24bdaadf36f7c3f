#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r4k; mkdir -p $O
export PYTHONPATH=$R
for sb in 1 2 4; do
  CDA_PPO_SUB_BATCHES=$sb timeout 300 python -m gym_continuousdoubleauction_amd.ppo --markets 4096 --agents 4 --horizon 64 --iters 8 --out $O/ppo_sb$sb.json > $O/ppo_sb$sb.log 2>&1
  python - <<PY
import json
d=json.load(open("$O/ppo_sb$sb.json")); h=d["iterations"][2:]
print("sub_batches=$sb e2e %.1f M agent-steps/s; rollout %.2f ms update %.2f ms" % (d["value"]/1e6, 1e3*sum(x["rollout_s"] for x in h)/len(h), 1e3*sum(x["update_s"] for x in h)/len(h)))
PY
done
timeout 300 python -m pytest tests/test_hip_mlp.py -q -m gpu 2>&1 | tail -1
python tools/mlp_timing.py --mt 1 --sample --rows 1024 --block 3 2>&1 | tail -18
