#!/bin/bash
# Run ON THE GPU BOX (round 6, first measurement call): the short policy leg taken apart + the league rollout by market count.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python tools/policy_leg_probe.py > gpurun_out/policy_leg_probe.jsonl 2> gpurun_out/policy_leg_probe.err
for n in 2048 4096 8192; do
  python -m gym_continuousdoubleauction_amd.league_train --fused --markets $n --agents 8 --episode 64 --iters 8 --out gpurun_out/league_${n}x8.json > gpurun_out/league_${n}x8.log 2>&1
done
