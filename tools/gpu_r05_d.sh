#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05; mkdir -p $O
for v in 0 1; do
  CDA_LEAGUE_UPDATE_STREAMS=$v timeout 600 python -m gym_continuousdoubleauction_amd.league_train --fused --markets 2048 --agents 8 --trainable 2 --episode 64 --iters 12 --out $O/league_us$v.json > /dev/null 2>&1
done
python - <<'PY'
import json
for n in ("league_us0", "league_us1"):
    try:
        d = json.load(open(f"gpurun_out/r05/{n}.json")); it = d["iterations"]
        print(n, round(d["value"] / 1e6, 1), "M;  rollout ms", [round(h["rollout_s"] * 1e3, 2) for h in it], " update ms", [round(h["update_s"] * 1e3, 2) for h in it])
    except Exception as e:
        print(n, "missing", e)
PY
timeout 300 python -m pytest tests/test_hip_league.py -q -m gpu -p no:cacheprovider -k "fused_league_training" 2>&1 | tail -3
