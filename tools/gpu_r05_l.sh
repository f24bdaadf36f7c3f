#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r05
for i in 1 2 3; do timeout 300 python tools/fb_wgrad_fusion_probe.py 2>&1 | grep "as built"; done
timeout 300 python tools/fb_wgrad_fusion_probe.py 2>&1 | grep "as built"
timeout 300 python -m gym_continuousdoubleauction_amd.ppo --iters 10 --out gpurun_out/r05/ppo_l.json > /dev/null 2>&1
timeout 300 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 12 --out gpurun_out/r05/league_l.json > /dev/null 2>&1
python - <<'PY'
import json, statistics
for n in ("ppo_l", "league_l"):
    d = json.load(open(f"gpurun_out/r05/{n}.json")); it = d["iterations"][2:]
    print(n, round(d["value"] / 1e6, 1), "M; rollout", round(statistics.median(x["rollout_s"] for x in it) * 1e3, 2), "update", round(statistics.median(x["update_s"] for x in it) * 1e3, 2))
PY
